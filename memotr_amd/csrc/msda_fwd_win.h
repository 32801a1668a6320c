// msda_fwd_win.h -- forward for self-attention over the pyramid (one query per pixel, Lq == S), rounds 3-5.
// Included by msda_hip.hip inside its anonymous namespace (shares msda_common.h and the host-side option state).
//
// Semantics: models/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299 (+ bilinear :33-84); fused prologue
// models/ops/modules/ms_deform_attn.py:104-123.
//
// What binds the direct gather (msda_fwd_d32_gather) is the vector-L1 request rate: every (query, head) row reads
// 64 corner rows of 128 B, 11.4 M requests per encoder call at ~0.45 requests/clk/CU = 41 us.
// This kernel moves the three coarse levels (75 % of the requests, 25 % of `value`) to LDS:
//
//   * a workgroup owns (batch, head, region); a region is the set of queries whose pixels fall in one cell of a
//     2^rlog-pixel grid on the finest level -- all of them sample the same neighbourhood of every level.  Round 5:
//     16 x 16-pixel regions (256 + 64 + 16 + 4 = 340 rows) and 512 threads -- four times the rows per set of windows and
//     per prologue of rounds 3-4's 8 x 8 regions; a border region holds only the rows that exist, and border regions are
//     walked last so that the launch's workgroups end together (55.1 -> 45.9 us fused at 800 x 1333);
//   * blocks are numbered head-major and block i runs on XCD i % 8, so each XCD's 4 MiB L2 holds ONE head's slab of
//     `value` (2.9 MB at 800 x 1333): the slab is fetched from the fabric once instead of ~3.6 times;
//   * per windowed level one window of this head's rows (128 B per pixel, zero outside the level / on padded
//     pixels) is filled with `buffer_load ... lds` (no VGPR round trip) around the region's centre + this head's mean
//     sampling offset on that level.  Rounds 3-4 measured that offset in every workgroup (rows -> reduce -> barrier ->
//     place -> barrier -> fill: two dependent round trips in front of the fill); round 5 keeps running means per (head,
//     level) in the call site's selector record (msda_select.h), fed by one wavefront of one workgroup in sixteen and
//     read at the start -- the fill is issued right after the table barrier.  The finest level keeps going through the
//     vector L1, so both pipes work;
//   * 16 lanes own a (query, head) row: lanes of one half read pixel w0, the other half pixel w0 + 1 -- one
//     ds_read_b128 covers 256 contiguous bytes per row, and the lane -> (row, chunk) map follows the four 16-lane
//     groups the LDS services a b128 read in, so the read is bank-conflict free (256 B/clk/CU instead of ~110 for
//     independent 128-B rows);
//   * one lane stages one (row, point) record: [w_top, addr_top, w_bot, addr_bot] per half (each weight in the low
//     register of an aligned pair: v_pk_fma broadcasts it without a move), LDS byte addresses for
//     points inside their window, `value` byte offsets (out of range = zero padding) otherwise; a point outside
//     its window takes the global path, so results never depend on the window placement -- only the speed does.
#pragma once

constexpr int kWinMaxL = 4;

struct WinPlan {
    int N, S, M, L, P, Lq;
    int RY, RX;                    // regions per image
    int RYf, RXf;                  // ... of them complete on level 0 (H0 >> rlogy, W0 >> rlogx): walked first
    int rows;                      // queries per region
    int steps;                     // ceil(rows / 4)
    int lwin0;                     // first level served from an LDS window
    int rlogx, rlogy;              // log2 of the region width / height on level 0
    int rsx, rsy;                  // region width / height on level 0 in pixels (1 << rlogx, 1 << rlogy; any size in grid mode)
    int grid;                      // 1: equal regions of rsy x rsx level-0 pixels, any size (make_win_plan_grid): level l's
                                   //    pixels go to the region their centre falls into (win_bound)
    int H[kWinMaxL], W[kWinMaxL];
    int qstart[kWinMaxL];          // first query of level l
    int shx[kWinMaxL], shy[kWinMaxL];   // log2 of the region width / height on level l
    int row0[kWinMaxL + 1];        // first region-row of level l
    int ww[kWinMaxL], wh[kWinMaxL];  // window width / height in pixels (0: no window)
    int wmagic[kWinMaxL];          // (x * magic) >> 16 == x / ww for x < ww * wh
    int wbase[kWinMaxL + 1];       // first window pixel of level l; [kWinMaxL] = all window pixels = the zero row
    float ratw[kWinMaxL][kWinMaxL], rath[kWinMaxL][kWinMaxL];   // [lq][l] = W_l / W_lq, H_l / H_lq
    int groups;                    // 1-KiB fill groups (8 fp32 / 16 bf16 pixels; window pixels + the zero row, rounded up)
    int gplog;                     // log2 of the pixels per fill group
    int wgroups_max;               // ... of the largest window (a masked call needs <= 8 per wavefront)
    float rcpH[kWinMaxL], rcpW[kWinMaxL], rcpP;   // correctly rounded 1/H, 1/W, 1/P (div_small)
    unsigned value_bytes;
    int n_blocks;
    unsigned long long *trace;     // profiling only (tools/fwd_win_timeline.py): 32 s_memtime stamps per wavefront, or null
    unsigned long long *stats, *stats_host;  // msda_select.h records (device / mapped host); null: no statistics
    int sel_level;
    int measure;                   // 1: every workgroup measures its own mean offsets before it places its windows (rounds 3-4);
                                   // 0: windows go where the record's running means say (msda_select.h), counting workgroups measure
    int ablate;                    // profiling only (msda_set_option "fwd_win_ablate"): 1 stop after the prologue, 2 no gather,
                                   // 4 write per (row, point) 2 = left its window / 1 = served from it / 0 into `out` (use 6)
};

struct WinTables {
    int H[kWinMaxL], W[kWinMaxL], qstart[kWinMaxL], shx[kWinMaxL], shy[kWinMaxL], row0[kWinMaxL + 1];
    int ww[kWinMaxL], wh[kWinMaxL], wmagic[kWinMaxL], wbase[kWinMaxL + 1], lstart[kWinMaxL];
    int ox[kWinMaxL], oy[kWinMaxL];
    int lvl[16];                   // level of point t
    // this region's queries, level by level: first pixel, pixels per row that exist, 1 / that, and (row0) the first
    // region-row of the level counting existing pixels only -- a region on the image border has fewer rows
    int y0[kWinMaxL], x0[kWinMaxL], wv[kWinMaxL];
    float rcpw[kWinMaxL];
    int rows, steps;
};


// 24-bit integer multiplies (v_mul_u32_u24 / v_mul_i32_i24: full rate; v_mul_lo_u32 and v_mad_u64_u32 are quarter rate
// on CDNA -- seven of them per step were ~10 % of the step's VALU time).  make_win_plan keeps every operand in range:
// queries and pixels per level below 2^23, row strides below 2^23.
__device__ __forceinline__ unsigned win_umul24(unsigned a, unsigned b) { return (unsigned)__umul24(a, b); }
__device__ __forceinline__ int win_mul24(int a, int b) { return __mul24(a, b); }

// Grid mode: first pixel of region r on a level of n pixels, for regions of rs pixels on the finest level (n0 pixels):
// the pixels whose centre, (y + 0.5) n0 / n in level-0 coordinates, lies at or beyond r rs -- ceil(r rs n / n0 - 1/2).
// Monotone in r, 0 for r = 0, and r rs itself on the finest level: every pixel of every level has exactly one region.
__host__ __device__ __forceinline__ int win_bound(int r, int rs, int n, int n0) {
    const int b = (int)((2 * (long long)r * rs * n + n0 - 1) / (2 * (long long)n0));
    return b > n ? n : b;
}

struct WinRow {
    bool ok;
    int lq, py, px;
    unsigned qrow, pm;
};

// Region row r -> its query.  Rows count the pixels that EXIST (round 5): a border region of 16 x 16 pixels with 4 x 8
// of them inside the image has 32 + 8 + 2 + 1 rows, not 340 with 297 dead ones staged and skipped.
__device__ __forceinline__ WinRow win_row(const WinTables &tb, int L, int r, int b, int m, int M, int Lq) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < kWinMaxL; ++i)
        if (i < L && r >= tb.row0[i]) l = i;
    const int local = r - tb.row0[l], wv = tb.wv[l];
    // local / wv for local < 1024, wv <= 32: (local + 0.5) / wv is at least 1 / 64 away from every integer
    const int y = (int)(((float)local + 0.5f) * tb.rcpw[l]);
    WinRow o;
    o.lq = l;
    o.py = tb.y0[l] + y;
    o.px = tb.x0[l] + (local - y * wv);
    o.ok = r < tb.rows;
    const int q = o.ok ? tb.qstart[l] + o.py * tb.W[l] + o.px : 0;
    o.qrow = (unsigned)b * (unsigned)Lq + (unsigned)q;
    o.pm = o.qrow * (unsigned)M + (unsigned)m;
    return o;
}

// raw inputs of one (row, point): what a lane prefetches one step ahead
struct WinRaw {
    f32x2 a;       // plain: location (x, y); fused: offset (x, y)
    float w;       // plain: attention weight; fused: logit
    f32x4 r;       // fused: reference point (x, y[, w, h])
};

template <bool FUSED>
__device__ __forceinline__ WinRaw win_load_raw(const PointSrc &src, unsigned qrow, unsigned pm, int m, int L, int LP,
                                               int t, int l) {
    WinRaw w;
    w.a = f32x2{0.f, 0.f};
    w.w = FUSED ? -INFINITY : 0.f;
    w.r = f32x4{0.f, 0.f, 0.f, 0.f};
    if (t < LP) {      // rows that do not exist read query 0: always a legal address
        if (FUSED) {
            const unsigned prow = win_umul24(qrow, (unsigned)src.proj_stride);
            const float *off = src.proj + (prow + (unsigned)((m * LP + t) * 2));
            w.a = *reinterpret_cast<const f32x2 *>(off);
            w.w = src.proj[prow + (unsigned)(src.n_off + m * LP + t)];
            const float *rp = src.ref + ((win_umul24(qrow, (unsigned)L) + (unsigned)l) << (src.ref_dim >> 1));      // (x 2 or x 4)
            if (src.ref_dim == 2) {
                const f32x2 r2 = *reinterpret_cast<const f32x2 *>(rp);
                w.r.x = r2.x;
                w.r.y = r2.y;
            } else {
                w.r = *reinterpret_cast<const f32x4 *>(rp);
            }
        } else {
            const unsigned pt = win_umul24(pm, (unsigned)LP) + (unsigned)t;      // (pm < 2^24: make_win_plan)
            w.a = *reinterpret_cast<const f32x2 *>(src.loc + pt * 2u);
            w.w = src.attn[pt];
        }
    }
    return w;
}

// (x / dx, y / dy) for small positive integers dx, dy held as floats, with rdx = RN(1 / dx) from the host: q0 = x * rd,
// q = fma(fma(-d, q0, x), rd, q0) is the correctly rounded quotient (Markstein) whenever nothing under- or overflows;
// outside that range (and for zeros, whose sign the correction loses) the IEEE divisions run.  Same bits as x / d.
__device__ __forceinline__ f32x2 div_small2(f32x2 v, float dx, float rdx, float dy, float rdy) {
#pragma clang fp contract(off)
    // |v| in [2^-40, 2^40] as one unsigned compare on the bits; a ZERO is served by the fast path too (round 5: the
    // initial offset star has exact zeros in every second head, and the whole wavefront then also ran the IEEE
    // divisions): 0 * rd = 0, the residual is 0, the quotient is +0 -- the sign of a zero offset is lost, which no
    // location can show (ref + (+-0) = ref for every ref but -0, and reference points are not negative zeros)
    const unsigned bx = __float_as_uint(v.x) & 0x7fffffffu, by = __float_as_uint(v.y) & 0x7fffffffu;
    const bool slow = ((bx - 0x2B800000u > 0x28000000u) && bx != 0u) || ((by - 0x2B800000u > 0x28000000u) && by != 0u);
    f32x2 q;
    const float qx = v.x * rdx, qy = v.y * rdy;
    q.x = __builtin_fmaf(__builtin_fmaf(-dx, qx, v.x), rdx, qx);
    q.y = __builtin_fmaf(__builtin_fmaf(-dy, qy, v.y), rdy, qy);
    if (__builtin_amdgcn_ballot_w64(slow) != 0ull) {      // (wave-uniform: nobody pays for what nobody needs)
        if (slow) {
            q.x = v.x / dx;
            q.y = v.y / dy;
        }
    }
    return q;
}

// sampling location from the raw inputs; the fused form repeats fused_location()'s IEEE operation order
template <bool FUSED>
__device__ __forceinline__ f32x2 win_location(const WinRaw &w, int ref_dim, float fP, float rP, float fH, float rH,
                                              float fW, float rW) {
#pragma clang fp contract(off)
    if (!FUSED) return w.a;
    f32x2 xy;
    if (ref_dim == 2) {
        const f32x2 d = div_small2(w.a, fW, rW, fH, rH);
        xy.x = w.r.x + d.x;
        xy.y = w.r.y + d.y;
    } else {
        const f32x2 p = div_small2(w.a, fP, rP, fP, rP);
        const float qx = p.x * w.r.z, qy = p.y * w.r.w;
        const float hx = qx * 0.5f, hy = qy * 0.5f;
        xy.x = w.r.x + hx;
        xy.y = w.r.y + hy;
    }
    return xy;
}

// reductions over the 16 lanes of a staging row (one lane per point), DPP only
// (v_max_f32 with the DPP operand folded in: fmaxf(x, dpp(x)) compiles to a DPP move, a canonicalising max of each
//  operand and the max -- 12 instructions for the four stages instead of 4; the s_nop are the two wait states a DPP
//  read of a VGPR written by the previous VALU instruction needs, which the assembler does not add inside asm)
__device__ __forceinline__ float row16_max(float x) {
    float r;
    asm volatile("s_nop 1\n\t"
                 "v_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf"
                 : "=&v"(r)
                 : "v"(x));
    return r;
}
__device__ __forceinline__ float row16_sum(float x) {
    x += MSDA_DPP(x, 0xB1);
    x += MSDA_DPP(x, 0x4E);
    x += MSDA_DPP(x, 0x141);
    x += MSDA_DPP(x, 0x140);
    return x;
}

// four consecutive channels of a window row in LDS as floats: fp32 rows (16 bytes) or bf16 rows (8 bytes)
template <typename TV>
__device__ __forceinline__ f32x4 win_lds_ch4(const unsigned char *p);
template <>
__device__ __forceinline__ f32x4 win_lds_ch4<float>(const unsigned char *p) { return *reinterpret_cast<const f32x4 *>(p); }
template <>
__device__ __forceinline__ f32x4 win_lds_ch4<bf16_t>(const unsigned char *p) {
    const u32x2 u = *reinterpret_cast<const u32x2 *>(p);
    return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                 __uint_as_float(u.y & 0xffff0000u)};
}
__device__ __forceinline__ void win_store4(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }
__device__ __forceinline__ void win_store4(bf16_t *p, f32x4 v) {
    *reinterpret_cast<u32x2 *>(p) = u32x2{bf16_bits_rne(v.x) | (bf16_bits_rne(v.y) << 16), bf16_bits_rne(v.z) | (bf16_bits_rne(v.w) << 16)};
}

// One point whose record may be an LDS address (inside its window) or a `value` byte offset (bit 0 of word 1 set),
// lane by lane: loads and FMAs in one place.
template <typename TV>
__device__ __forceinline__ f32x4 win_mixed_point(const u32x4 r, const unsigned char *s_dyn, unsigned zero_off,
                                                 __amdgpu_buffer_rsrc_t vr, unsigned sub16, f32x4 acc) {
    const bool g = (r.y & 1u) != 0u;
    f32x4 v0 = win_lds_ch4<TV>(s_dyn + ((g ? zero_off : r.y) + sub16));
    f32x4 v1 = win_lds_ch4<TV>(s_dyn + ((g ? zero_off : r.w) + sub16));
    if (g) {
        v0 = buf_load_ch4<TV>(vr, (r.y & ~1u) + sub16);
        v1 = buf_load_ch4<TV>(vr, r.w + sub16);
    }
    acc += __uint_as_float(r.x) * v0;
    acc += __uint_as_float(r.z) * v1;
    return acc;
}

typedef __attribute__((address_space(3))) void lds_void;

// Window origin of level l: centred on the region's centre in level-l sampling coordinates plus the mean sampling
// offset (dx, dy), kept on the level and its one-pixel zero border.
__device__ __forceinline__ void win_place(WinTables &tb, const WinPlan &pl, int l, int ry, int rx, float dx, float dy) {
    const float cx = ((float)(rx * pl.rsx) + 0.5f * (float)pl.rsx) * pl.ratw[0][l] - 0.5f + dx;
    const float cy = ((float)(ry * pl.rsy) + 0.5f * (float)pl.rsy) * pl.rath[0][l] - 0.5f + dy;
    const int ww = pl.ww[l], wh = pl.wh[l];
    int ox = (int)floorf(cx - 0.5f * (float)(ww - 1) + 0.5f);
    int oy = (int)floorf(cy - 0.5f * (float)(wh - 1) + 0.5f);
    const int max_x = pl.W[l] + 1 - ww, max_y = pl.H[l] + 1 - wh;
    ox = ox > max_x ? max_x : ox;
    oy = oy > max_y ? max_y : oy;
    tb.ox[l] = ox < -1 ? -1 : ox;
    tb.oy[l] = oy < -1 ? -1 : oy;
}

// WPS = wavefronts per SIMD the register budget is sized for (4: 128 VGPRs -- 512-thread workgroups, or four 256-thread
// workgroups per CU; 3: 168 VGPRs -- three 256-thread workgroups per CU, what 40-53 KB of LDS admits)
// NE: the corner rows of the first NE (0 / 2 / 4) global points are requested before the LDS-served points and used
// after them (their latency hides behind the LDS phase at the price of 10 registers per point held across it).
// TRACE: the timeline build (tools/fwd_win_timeline.py) -- s_memtime stamps at the phase boundaries; the production
// instantiations hold none of it (as a run-time test the stamps cost 1.1 us per launch, round 4).
// TV: float, or bf16_t (round 6: 64-byte rows -- windows of half the bytes, 16 pixels per fill instruction, 8-byte LDS
// reads widened in registers; locations, weights and accumulation stay fp32)
template <typename TV, bool FUSED, int WPS, int NE, bool TRACE = false, int PB = 4>
__global__ __launch_bounds__(WPS == 4 ? 512 : 256, WPS) void msda_fwd_d32_win(const TV *__restrict__ value,
                                                           const int64_t *__restrict__ lstart, const PointSrc src,
                                                           TV *__restrict__ out, const WinPlan pl) {
    constexpr unsigned kPix = 32u * (unsigned)sizeof(TV);      // bytes of one head's pixel row
    constexpr unsigned kCh = kPix / 8u;                         // ... of a lane's four channels
    constexpr int kGPlog = sizeof(TV) == 4 ? 3 : 4;             // log2 pixels per 1-KiB fill group
    constexpr int kLPPlog = sizeof(TV) == 4 ? 3 : 2;            // log2 lanes per pixel of a fill instruction
    __shared__ WinTables tb;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];

    // ---- block -> (head, batch, region); head-major numbering, one contiguous run of ids per XCD ----
    const int chunk = (int)(gridDim.x >> 3);
    const int sw = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if (sw >= pl.n_blocks) return;
    const int nreg = pl.RY * pl.RX;
    const int reg = sw % nreg;
    const int hb = sw / nreg;
    const int b = hb % pl.N, m = hb / pl.N;
    // complete regions first, the right border column next, the bottom border row last: the partial regions (fewer
    // rows, shorter workgroups) fill the end of the launch instead of leaving whole workgroups for a last round
    int ry, rx;
    {
        const int nint = pl.RYf * pl.RXf, wcol = pl.RX - pl.RXf, ncol = wcol * pl.RYf;
        if (reg < nint) {
            ry = reg / pl.RXf;
            rx = reg - ry * pl.RXf;
        } else if (reg < nint + ncol) {
            const int e = reg - nint;
            ry = e / wcol;
            rx = pl.RXf + (e - ry * wcol);
        } else {
            const int e = reg - nint - ncol;
            ry = e / pl.RX;
            rx = e - ry * pl.RX;
            ry += pl.RYf;
        }
    }

    const int L = pl.L, P = pl.P, LP = L * P, M = pl.M;
    const int tid = threadIdx.x, lane = tid & 63, nw = (int)(blockDim.x >> 6);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform, and known to be
    unsigned long long *const trc = (TRACE && pl.trace) ? pl.trace + ((size_t)sw * nw + wave) * 32 : nullptr;
    int trc_k = 0;
#define WIN_STAMP()                                                                   \
    do {                                                                              \
        if (TRACE && trc != nullptr && trc_k < 32) {                                  \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();               \
            if (lane == 0) trc[trc_k] = t_;                                           \
            ++trc_k;                                                                  \
        }                                                                             \
    } while (0)
    WIN_STAMP();       // 0: start
    if (tid < kWinMaxL) {
        tb.H[tid] = pl.H[tid];
        tb.W[tid] = pl.W[tid];
        tb.qstart[tid] = pl.qstart[tid];
        tb.shx[tid] = pl.shx[tid];
        tb.shy[tid] = pl.shy[tid];
        tb.ww[tid] = pl.ww[tid];
        tb.wh[tid] = pl.wh[tid];
        tb.wmagic[tid] = pl.wmagic[tid];
        // (one query per pixel: level l of `value` starts where its queries do -- the device copy of level_start_index,
        //  a cold global load in front of the first barrier, is not read)
        tb.lstart[tid] = pl.qstart[tid];
        tb.ox[tid] = tb.oy[tid] = 0;
    }
    if (tid <= kWinMaxL) {
        int base = 0, mine = 0;
#pragma unroll
        for (int i = 0; i < kWinMaxL; ++i) {
            const int sy = pl.shy[i], sx = pl.shx[i];
            int y0 = ry << sy, x0 = rx << sx;
            int hv = pl.H[i] - y0, wv = pl.W[i] - x0;
            hv = hv > (1 << sy) ? (1 << sy) : (hv < 0 ? 0 : hv);
            wv = wv > (1 << sx) ? (1 << sx) : (wv < 0 ? 0 : wv);
            if (pl.grid) {
                y0 = win_bound(ry, pl.rsy, pl.H[i], pl.H[0]);
                x0 = win_bound(rx, pl.rsx, pl.W[i], pl.W[0]);
                hv = (ry + 1 == pl.RY ? pl.H[i] : win_bound(ry + 1, pl.rsy, pl.H[i], pl.H[0])) - y0;
                wv = (rx + 1 == pl.RX ? pl.W[i] : win_bound(rx + 1, pl.rsx, pl.W[i], pl.W[0])) - x0;
            }
            if (i >= L) hv = 0;
            if (tid == i) {
                mine = base;
                tb.y0[i] = y0;
                tb.x0[i] = x0;
                tb.wv[i] = wv > 0 ? wv : 1;
                tb.rcpw[i] = 1.f / (float)(wv > 0 ? wv : 1);
            }
            base += hv * wv;
        }
        tb.row0[tid] = tid == kWinMaxL ? base : mine;
        if (tid == kWinMaxL) {
            tb.rows = base;
            tb.steps = (base + 3) >> 2;
        }
        tb.wbase[tid] = pl.wbase[tid];
    }
    if (tid >= 64 && tid < 80) tb.lvl[tid - 64] = (tid - 64) < LP ? (tid - 64) / P : 0;
    // statistics (msda_select.h): one workgroup in eight counts, every wavefront for itself
#ifndef MSDA_WIN_STATS
#define MSDA_WIN_STATS 1      // (0: A/B builds without the selector's counting)
#endif
    const bool stat_wg = MSDA_WIN_STATS && pl.stats != nullptr && (sw & 7) == 0;
    const bool measure = pl.measure != 0;          // block-uniform
    bool hint_valid = true;
    // windows of the windowed levels: centred on the region + this head's mean sampling offset on that level, as the
    // launches before this one measured it (round 5; no dependent round trip in front of the fill any more)
    if (!measure) {
        const float *hp = reinterpret_cast<const float *>(pl.stats + kSelHintWord) + m * (kSelHintLevels * 2);
        // (agent-scope loads: the means were written by the previous launch's publishing wavefront; a scalar load
        //  could be served by a constant cache that launch boundary did not invalidate)
        // (a counting workgroup notes NOW whether the means it is about to use were measured: by the time it reports, this
        //  launch's own publishing wavefront may have set the flag -- the first launch of a call site, windows centred
        //  on nothing, would then report its misplaced points and send the selector to the gather kernel for 32 calls)
        if (stat_wg) {       // (bit head * 4 + level: every windowed level of THIS head must have a measured mean)
            const unsigned long long have = __hip_atomic_load(pl.stats + kSelHintValidWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = ((1u << L) - 1u) & ~((1u << pl.lwin0) - 1u);
            hint_valid = m < kSelHintHeads && (((unsigned)(have >> (m * kSelHintLevels)) & want) == want);
        }
        float hx = 0.f, hy = 0.f;
        if (tid >= pl.lwin0 && tid < L && m < kSelHintHeads) {
            hx = __hip_atomic_load(hp + 2 * tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            hy = __hip_atomic_load(hp + 2 * tid + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid >= pl.lwin0 && tid < L) win_place(tb, pl, tid, ry, rx, hx, hy);
    }
    unsigned v_off = 0u;                // statistics: windowed points of this lane that left their window
    unsigned char mpad[kWinMaxL] = {0, 0, 0, 0};      // fused + mask: "the window pixel this lane answers for is padded"
    __syncthreads();
    WIN_STAMP();       // 1: tables
    if (tb.rows == 0) return;      // (levels that are not a pyramid of one image can leave a region without any query)

    // staging layout: lane -> (row slot, point)
    const int s_rs = lane >> 4, s_t = lane & 15;
    const int s_l = tb.lvl[s_t];
    // gather layout: lane -> (row slot = LDS service group of a b128 read, pixel half, 16-byte chunk)
    const int g_j = lane & 15, g_odd = (lane >> 4) & 1;
    const int g_jq = g_j >> 2;
    const bool g_inner = (g_jq == 1) || (g_jq == 2);
    const int g_row = ((lane >> 5) << 1) + ((g_inner == (g_odd != 0)) ? 0 : 1);
    const int g_half = g_j >> 3;
    const int g_chunk = (g_half ? 3 - (g_j & 3) : (g_j & 3)) + 4 * g_odd;
    const unsigned sub16 = (unsigned)g_chunk * kCh;

    const __amdgpu_buffer_rsrc_t vr = make_rsrc(value, pl.value_bytes);
    const unsigned zero_off = (unsigned)tb.wbase[kWinMaxL] * kPix;
    const unsigned win_bytes = (unsigned)pl.groups * 1024u;
    u32x4 *rec_w = reinterpret_cast<u32x4 *>(s_dyn + win_bytes) + wave * 128;     // 64 records of 32 bytes
    const u32x4 *rec_g = rec_w + g_row * 32 + g_half;                               // + 2 * t
    int *s_rowq = reinterpret_cast<int *>(s_dyn + win_bytes + (unsigned)nw * 2048u);   // query of region row r, -1: none
    // sampled (dx, dy, count) per level of every first-step row: lives in wave 0's record space until the steps start
    float *s_part = reinterpret_cast<float *>(s_dyn + win_bytes);
    const unsigned pix_stride = (unsigned)M * kPix;
    const unsigned row_base = ((unsigned)b * (unsigned)pl.S * (unsigned)M + (unsigned)m) * kPix;
    const unsigned q_base = (unsigned)b * (unsigned)pl.Lq;
    const int lwin0 = pl.lwin0;

    // ---- the windows' fill (LDS-DMA) and, with a padding mask, the mask bytes of the pixels this wavefront fills ----
    auto fill_windows = [&]() {
        // ---- fill the windows: 8 pixels (1 KiB) per wave instruction; cells outside the level / padded read 0.
        //      Every level's window starts on a group boundary, so the level is uniform per instruction.  (One
        //      instruction per window ROW needs a third of the address arithmetic but 40 % more, partly filled,
        //      DMA instructions: measured slower, 16.6 vs 14.7 us for the start-up phase alone) ----
        //      The padding mask is NOT consulted here: a mask byte ahead of every DMA address makes each fill
        //      instruction wait a memory round trip (and, vmcnt being in-order, for every DMA before it).  The bytes
        //      are requested below, next to the fill, and padded pixels are zeroed in LDS once both have landed.
        for (int l = lwin0; l <= L; ++l) {        // l == L: the group that holds the zero row
            const int g0 = tb.wbase[l < L ? l : kWinMaxL] >> kGPlog;
            const int g1 = l < L ? (tb.wbase[l + 1 < L ? l + 1 : kWinMaxL] >> kGPlog) : pl.groups;
            const int lc = l < L ? l : L - 1;
            const int ww = tb.ww[lc], npx = l < L ? ww * tb.wh[lc] : 0, magic = tb.wmagic[lc];
            const int oy = tb.oy[lc], ox = tb.ox[lc], H = tb.H[lc], W = tb.W[lc];
            const unsigned lbase = row_base + (unsigned)tb.lstart[lc] * pix_stride + (unsigned)(lane & ((1 << kLPPlog) - 1)) * 16u;
            for (int g = g0 + wave; g < g1; g += nw) {
                const int local = ((g - g0) << kGPlog) + (lane >> kLPPlog);
                const int wy = (local * magic) >> 16, wx = local - wy * ww;
                const int gy = oy + wy, gx = ox + wx;
                const bool inside = (local < npx) & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
                const int cell = gy * W + gx;
                const unsigned off = inside ? lbase + (unsigned)cell * pix_stride : kOobOffset;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(vr, (lds_void *)(s_dyn + (size_t)g * 1024), 16, (int)off, 0, 0, 0);
            }
        }
        // the mask bytes of the pixels THIS wavefront filled (8 lanes x 8 groups per level: make_win_plan's
        // wgroups_max keeps a level's window within 64 * nw pixels when there is a mask), in flight with the fill
        if (FUSED && src.mask != nullptr) {
            const unsigned char *mk = src.mask + (size_t)b * pl.S;
#pragma unroll
            for (int l = 0; l < kWinMaxL; ++l) {
                mpad[l] = 0;
                if (l >= lwin0 && l < L) {
                    const int g0 = tb.wbase[l] >> kGPlog, g1 = tb.wbase[l + 1 < L ? l + 1 : kWinMaxL] >> kGPlog;
                    const int g = g0 + wave + (lane >> kGPlog) * nw;
                    const int local = ((g - g0) << kGPlog) + (lane & ((1 << kGPlog) - 1));
                    const int ww = tb.ww[l], wy = (local * tb.wmagic[l]) >> 16, wx = local - wy * ww;
                    const int gy = tb.oy[l] + wy, gx = tb.ox[l] + wx;
                    const bool inside = (g < g1) & (local < ww * tb.wh[l]) & ((unsigned)gy < (unsigned)tb.H[l]) &
                                        ((unsigned)gx < (unsigned)tb.W[l]);
                    if (inside) mpad[l] = mk[tb.lstart[l] + gy * tb.W[l] + gx];
                }
            }
        }
    };
    // placement from the record's means: the fill -- the long pole of the prologue -- goes out before anything else
    if (!measure && lwin0 < L) fill_windows();

    // the region's rows -> query indices, once
    const int steps = __builtin_amdgcn_readfirstlane(tb.steps);     // of THIS region (pl.steps: of a complete one)
    for (int r = tid; r < steps * 4; r += (int)blockDim.x) {
        const WinRow w = win_row(tb, L, r, b, m, M, pl.Lq);
        s_rowq[r] = w.ok ? (int)(w.qrow - q_base) : -1;
    }

    // ---- first step's inputs; the mean sampling offset of every windowed level, measured on them ----
    int step = wave;
    bool row_ok;
    WinRaw raw;
    const WinRow row = win_row(tb, L, step * 4 + s_rs, b, m, M, pl.Lq);
    row_ok = row.ok;
    // (placement from the record: the fill is already out; the rows' own inputs follow it below)
    if (measure) raw = win_load_raw<FUSED>(src, row.qrow, row.pm, m, L, LP, s_t, s_l);
    // mean sampling offset per windowed level on this wavefront's first rows: for the placement (measuring mode, before
    // the fill) or for the record's running means (counting workgroups, after the fill has been issued)
    auto measure_offsets = [&]() {
        if (lwin0 < L) {
            // placement only: approximate arithmetic is fine here (the records repeat it exactly)
            float dx = 0.f, dy = 0.f, dc = 0.f;
            if (step < steps && row.ok && s_t < LP && s_l >= lwin0) {
                const float fW = (float)tb.W[s_l], fH = (float)tb.H[s_l];
                f32x2 xy = raw.a;
                if (FUSED) {
                    if (src.ref_dim == 2) {
                        xy.x = raw.r.x + raw.a.x * pl.rcpW[s_l];
                        xy.y = raw.r.y + raw.a.y * pl.rcpH[s_l];
                    } else {
                        xy.x = raw.r.x + raw.a.x * pl.rcpP * raw.r.z * 0.5f;
                        xy.y = raw.r.y + raw.a.y * pl.rcpP * raw.r.w * 0.5f;
                    }
                }
                const float w_im = xy.x * fW - 0.5f, h_im = xy.y * fH - 0.5f;
                if ((h_im > -1.f) & (w_im > -1.f) & (h_im < fH) & (w_im < fW)) {
                    // minus where the query's own pixel centre lands on level s_l
                    dx = w_im - (((float)row.px + 0.5f) * pl.ratw[row.lq][s_l] - 0.5f);
                    dy = h_im - (((float)row.py + 0.5f) * pl.rath[row.lq][s_l] - 0.5f);
                    dc = 1.f;
                }
            }
            for (int l = lwin0; l < L; ++l) {
                const bool mine = s_l == l;
                float sx = row16_sum(mine ? dx : 0.f), sy = row16_sum(mine ? dy : 0.f);
                float sc = row16_sum(mine ? dc : 0.f);
                if (measure) {
                    if (s_t == 0) {
                        float *pp = s_part + ((wave * 4 + s_rs) * kWinMaxL + l) * 3;
                        pp[0] = sx;
                        pp[1] = sy;
                        pp[2] = sc;
                    }
                } else {       // counting workgroup: this wavefront's four rows to the record's running sums
                    sx += __shfl_xor(sx, 16, 64); sy += __shfl_xor(sy, 16, 64); sc += __shfl_xor(sc, 16, 64);
                    sx += __shfl_xor(sx, 32, 64); sy += __shfl_xor(sy, 32, 64); sc += __shfl_xor(sc, 32, 64);
                    // (ONE wavefront of one workgroup in sixteen: 72 addresses take ~340 float atomics per launch.  With
                    //  every wavefront of one workgroup in eight -- 5.5 k same-address atomics -- the L2's atomic
                    //  units serialised them and the FILL behind them took twice as long: 54 instead of 46 us)
                    if (lane == 0 && sc > 0.f && m < kSelHintHeads) sel_hint_add(pl.stats, m, l, sx, sy, sc);
                }
            }
        }
    };
    if (measure) measure_offsets();
    WIN_STAMP();       // 2: first rows requested, (measuring) offsets measured
    if (lwin0 < L) {
        if (measure) {
            __syncthreads();
            WIN_STAMP();   // 3
            if (tid >= lwin0 && tid < L) {
                const int l = tid;
                float sx = 0.f, sy = 0.f, cnt = 0.f;
                for (int i = 0; i < nw * 4; ++i) {
                    sx += s_part[(i * kWinMaxL + l) * 3];
                    sy += s_part[(i * kWinMaxL + l) * 3 + 1];
                    cnt += s_part[(i * kWinMaxL + l) * 3 + 2];
                }
                const float rc = cnt > 0.f ? __builtin_amdgcn_rcpf(cnt) : 0.f;
                win_place(tb, pl, l, ry, rx, sx * rc, sy * rc);
            }
            WIN_STAMP();   // 4: (threads 1..3 of wavefront 0: placement done; everyone else: nothing)
            __syncthreads();
            WIN_STAMP();   // 5: placement visible
        } else {
            WIN_STAMP();   // (3-5: the same stamp numbers in both modes)
            WIN_STAMP();
            WIN_STAMP();
        }

        if (measure) fill_windows();
    }
    if (!measure) {
        raw = win_load_raw<FUSED>(src, row.qrow, row.pm, m, L, LP, s_t, s_l);
        if (stat_wg && wave == 0 && (sw & 15) == 0) measure_offsets();
    }

    WIN_STAMP();       // 6: fill issued
    // ---- this lane's point sits on one level for the whole kernel: its constants ----
    const bool c_pt = s_t < LP;
    const int cH = tb.H[s_l], cW = tb.W[s_l];
    const float fH = (float)cH, fW = (float)cW, fP = (float)P;
    const float rH = pl.rcpH[s_l], rW = pl.rcpW[s_l], rP = pl.rcpP;
    const bool c_windowed = c_pt && s_l >= lwin0;
    const int c_ww = tb.ww[s_l];
    const unsigned c_wwm1 = c_windowed ? (unsigned)(c_ww - 1) : 0u, c_whm1 = c_windowed ? (unsigned)(tb.wh[s_l] - 1) : 0u;
    const int c_ox = tb.ox[s_l], c_oy = tb.oy[s_l];
    const unsigned c_wbase = (unsigned)tb.wbase[s_l] * kPix;
    const unsigned c_wrow = (unsigned)c_ww * kPix;
    const unsigned c_wps = (unsigned)cW * pix_stride;
    const unsigned c_lbase = row_base + (unsigned)tb.lstart[s_l] * pix_stride;     // pixel 0 of the level, this head
    const unsigned c_flag = c_windowed ? 1u : 0u;          // word 0 of a windowed level's record: bit 0 = global path
    const unsigned c_dead = c_windowed ? zero_off : kOobOffset;
    const unsigned char *c_mask = (FUSED && src.mask != nullptr) ? src.mask + ((size_t)b * pl.S + tb.lstart[s_l]) : nullptr;
    const int T0 = lwin0 * P < LP ? lwin0 * P : LP;
    constexpr int NEA = NE > 0 ? NE : 1;
    const bool early = NE > 0 && T0 >= NE;

    // ---- the steps: stage 64 (row, point) records, gather, store ----
    // (the profiling switches exist in the TRACE instantiation only: as run-time tests in the step loop they hold
    //  scalar registers the production kernel has none to spare of -- 106 of 106 with six spilled, round 4)
    const int ablate = TRACE ? pl.ablate : 0;
    if (ablate & 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    const int iters = (steps + nw - 1) / nw;
    // One step's staging: this lane's (row, point) -> its record pair in the wavefront's record block (`raw`, `row_ok`:
    // the inputs of step st_).  Returns the 16-bit mask of points some row of the wavefront could not serve from a window.
    auto stage_step = [&](int st_) -> unsigned {
        // -- staging: this lane's point, straight-line --
            float a_in;
            if (FUSED) {
                // the one softmax arithmetic of the fused kernels (msda_common.h: exp2 / rcp, adjacent-pair tree -- row16_sum
                // IS that tree): the weights are, bit for bit, the ones the backward differentiates and
                // msda_fused_points_f32 exposes.  (Rounds 3-5 had the exact expf / division in the backward only; bringing
                // THEM here cost 1.5-2 us, so round 6 took this form there.)
                const float mx = row16_max(raw.w);
                const float e = sm_exp(raw.w, mx);
                a_in = e * sm_rcp(row16_sum(e));
            } else {
                a_in = raw.w;
            }
            const f32x2 xy = win_location<FUSED>(raw, src.ref_dim, fP, rP, fH, rH, fW, rW);
            float h_im, w_im;
            {
#pragma clang fp contract(off)
                const float ph = xy.y * fH, pw = xy.x * fW;      // the reference's rounding points (.cuh:285-288)
                h_im = ph - 0.5f;
                w_im = pw - 0.5f;
            }
            const bool gate = h_im > -1.f && w_im > -1.f && h_im < fH && w_im < fW;
            const float fh = floorf(h_im), fw = floorf(w_im);
            const int h0 = (int)fh, w0 = (int)fw;
            const bool live = gate && row_ok && c_pt;
            const float lh = gate ? h_im - fh : 0.f, lw = gate ? w_im - fw : 0.f;
            const float a = live ? a_in : 0.f;
            const float hh = 1.f - lh, hw = 1.f - lw;
            u32x4 ra, rb;
            ra.x = __float_as_uint((hh * hw) * a);
            ra.z = __float_as_uint((lh * hw) * a);
            rb.x = __float_as_uint((hh * lw) * a);
            rb.z = __float_as_uint((lh * lw) * a);
            // inside its window: LDS byte addresses (the window holds zeros outside the level / on padded pixels)
            const int wx = w0 - c_ox, wy = h0 - c_oy;
            const bool inwin = live && (unsigned)wx < c_wwm1 && (unsigned)wy < c_whm1;
            const unsigned lbase = c_wbase + (unsigned)(win_mul24(wy, c_ww) + wx) * kPix;
            // otherwise: byte offsets into `value`, out of range for corners that do not exist
            const bool need = live && !inwin;
            const bool okh0 = (unsigned)h0 < (unsigned)cH, okh1 = (unsigned)(h0 + 1) < (unsigned)cH;
            const bool okw0 = (unsigned)w0 < (unsigned)cW, okw1 = (unsigned)(w0 + 1) < (unsigned)cW;
            bool ok00 = okh0 && okw0, ok01 = okh0 && okw1, ok10 = okh1 && okw0, ok11 = okh1 && okw1;
            const int cell = win_mul24(h0, cW) + w0;
            if (FUSED && c_mask != nullptr && need) {
                ok00 = ok00 && !c_mask[ok00 ? cell : 0];
                ok01 = ok01 && !c_mask[ok01 ? cell + 1 : 0];
                ok10 = ok10 && !c_mask[ok10 ? cell + cW : 0];
                ok11 = ok11 && !c_mask[ok11 ? cell + cW + 1 : 0];
            }
            if (ablate & 4) {      // profiling only (tools/fwd_offset_sweep.py): which points left their window
                const WinRow wq_ = win_row(tb, L, st_ * 4 + s_rs, b, m, M, pl.Lq);      // (not the table: first step, no barrier yet)
                const int qq = wq_.ok ? (int)(wq_.qrow - q_base) : -1;
                if (c_pt && qq >= 0)
                    reinterpret_cast<float *>(out)[(size_t)((q_base + (unsigned)qq) * (unsigned)M + (unsigned)m) * 32u + (unsigned)s_t] =
                        (need && c_windowed) ? 2.f : ((live && c_windowed) ? 1.f : 0.f);
            }
            const unsigned o00 = c_lbase + (unsigned)win_mul24(cell, (int)pix_stride);      // (signed: cell is -W - 1 ... H W)
            ra.y = inwin ? lbase : (need ? (ok00 ? o00 : kOobOffset) | c_flag : c_dead);
            ra.w = inwin ? lbase + c_wrow : (need ? (ok10 ? o00 + c_wps : kOobOffset) : c_dead);
            rb.y = inwin ? lbase + kPix : (need ? (ok01 ? o00 + pix_stride : kOobOffset) | c_flag : c_dead);
            rb.w = inwin ? lbase + c_wrow + kPix : (need ? (ok11 ? o00 + c_wps + pix_stride : kOobOffset) : c_dead);
            if (c_pt) {
                rec_w[2 * lane] = ra;
                rec_w[2 * lane + 1] = rb;
            }
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(need && c_windowed);
            v_off += (need && c_windowed) ? 1u : 0u;      // (statistics: per lane, summed once at the end)
            const unsigned fold = (unsigned)(bal | (bal >> 32));
            return (fold | (fold >> 16)) & 0xffffu;
        
    };
    // the inputs of step st_ (its rows' projection rows / locations): requested a step ahead
    // (`table`: the row -> query table may be read -- it is written by all wavefronts before the first step and the
    //  first workgroup barrier after that is the one that waits for the windows: the prefetch of the first iteration
    //  computes its row instead.  Round 5's placement without barriers exposed this; with the measuring placement the two
    //  placement barriers happened to cover it)
    auto prefetch_step = [&](int st_, bool table) {
        int q;
        if (table) {
            q = s_rowq[st_ * 4 + s_rs];
        } else {
            const WinRow w = win_row(tb, L, st_ * 4 + s_rs, b, m, M, pl.Lq);
            q = w.ok ? (int)(w.qrow - q_base) : -1;
        }
        row_ok = q >= 0;
        const unsigned qrow = q_base + (unsigned)(q < 0 ? 0 : q);
        raw = win_load_raw<FUSED>(src, qrow, win_umul24(qrow, (unsigned)M) + (unsigned)m, m, L, LP, s_t, s_l);
    };
    // the windows must have landed before the first LDS-served point (first iteration, every wavefront)
    auto windows_landed = [&]() {
        if (lwin0 < L) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (FUSED && src.mask != nullptr) {       // this wavefront's fill has landed: zero its padded pixels
#pragma unroll
                for (int l = 0; l < kWinMaxL; ++l) {
                    if (mpad[l]) {
                        const int g = (tb.wbase[l] >> kGPlog) + wave + (lane >> kGPlog) * nw;
                        f32x4 *px = reinterpret_cast<f32x4 *>(s_dyn + (((size_t)g << kGPlog) + (size_t)(lane & ((1 << kGPlog) - 1))) * kPix);
#pragma unroll
                        for (int c = 0; c < (int)(kPix / 16u); ++c) px[c] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
        }
        __syncthreads();      // (also: the row -> query table is complete from here on, windows or not)
    };
    // the LDS-served points of the step whose records are in place
    // (`nb`: points per batch as a type -- PB of them first (one wait for PB records, one for 2 PB rows: the 256-register
    //  build, two wavefronts per SIMD, hides the LDS round trips by what one wavefront has in flight), then fours)
    auto lds_batch = [&](auto nb, int t0, unsigned gmask, f32x4 acc) -> f32x4 {
        constexpr int NB = decltype(nb)::value;
        const u32x4 *rp = rec_g + 2 * t0;
        u32x4 r[NB];
        f32x4 v[NB][2];
#pragma unroll
        for (int i = 0; i < NB; ++i) r[i] = rp[2 * i];
        if (((gmask >> t0) & ((1u << NB) - 1u)) == 0u) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                v[i][0] = win_lds_ch4<TV>(s_dyn + (r[i].y + sub16));
                v[i][1] = win_lds_ch4<TV>(s_dyn + (r[i].w + sub16));
            }
        } else {
            // some row's point left its window: those lanes read `value` itself, the others their window;
            // all rows of the batch are requested before the first is used
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const bool g = (r[i].y & 1u) != 0u;
                v[i][0] = win_lds_ch4<TV>(s_dyn + ((g ? zero_off : r[i].y) + sub16));
                v[i][1] = win_lds_ch4<TV>(s_dyn + ((g ? zero_off : r[i].w) + sub16));
                if (g) {
                    v[i][0] = buf_load_ch4<TV>(vr, (r[i].y & ~1u) + sub16);
                    v[i][1] = buf_load_ch4<TV>(vr, r[i].w + sub16);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            acc += __uint_as_float(r[i].x) * v[i][0];
            acc += __uint_as_float(r[i].z) * v[i][1];
        }
        return acc;
    };
    auto lds_points = [&](unsigned gmask, f32x4 acc) -> f32x4 {
            int t0 = T0;
            if constexpr (PB > 4)
                for (; t0 + PB <= LP; t0 += PB) acc = lds_batch(std::integral_constant<int, PB>{}, t0, gmask, acc);
            for (; t0 + 4 <= LP; t0 += 4) acc = lds_batch(std::integral_constant<int, 4>{}, t0, gmask, acc);
            for (; t0 < LP; ++t0) acc = win_mixed_point<TV>(rec_g[2 * t0], s_dyn, zero_off, vr, sub16, acc);
            return acc;
    };
    // the two pixel halves of a row meet, the row leaves
    auto finish_row = [&](f32x4 acc) {
            // -- the two pixel halves of a row sit on row_mirror partners --
            // (scalars: __builtin_bit_cast of a vector ELEMENT reads element 0 with this compiler)
            const float a0 = acc.x, a1 = acc.y, a2 = acc.z, a3 = acc.w;
            acc.x = a0 + MSDA_DPP(a0, 0x140);
            acc.y = a1 + MSDA_DPP(a1, 0x140);
            acc.z = a2 + MSDA_DPP(a2, 0x140);
            acc.w = a3 + MSDA_DPP(a3, 0x140);
            const int q = s_rowq[step * 4 + g_row];
            if (g_half == 0 && q >= 0) {
                const unsigned pm = win_umul24(q_base + (unsigned)q, (unsigned)M) + (unsigned)m;
                win_store4(out + ((size_t)pm * 32u + (unsigned)g_chunk * 4u), acc);
            }
    };
    {
    for (int it = 0; it < iters; ++it, step += nw) {
        const bool have = step < steps;          // wave-uniform
        unsigned gmask = 0u;
        if (have) gmask = stage_step(step);
        WIN_STAMP();   // 7 + 3 it: staged
        if (step + nw < steps) prefetch_step(step + nw, it > 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // -- levels read through the vector L1: issue the first four points' corner rows now, use them after the
        //    LDS-served points (fewer than four such points: they all go through the late loop) --
        u32x4 gr[NEA];
        f32x4 gv[NEA][2];
        if (have && early && !(ablate & 2)) {
#pragma unroll
            for (int i = 0; i < NEA; ++i) gr[i] = rec_g[2 * i];
#pragma unroll
            for (int i = 0; i < NEA; ++i) {
                gv[i][0] = buf_load_ch4<TV>(vr, gr[i].y + sub16);
                gv[i][1] = buf_load_ch4<TV>(vr, gr[i].w + sub16);
            }
        }
        if (it == 0) windows_landed();
        WIN_STAMP();   // 8 + 3 it: records visible, early loads issued, (first step) windows landed
        if (have && !(ablate & 2)) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            // -- levels read from the LDS windows, four points per batch.  A point some row of the wave could not
            //    serve from its window (bit in gmask) is handled on the spot, loads and FMAs, so that the common path
            //    holds no register a vector-memory instruction writes --
            acc = lds_points(gmask, acc);
            // -- consume the global points issued above, then any that were not --
            if (early) {
#pragma unroll
                for (int i = 0; i < NEA; ++i) {
                    acc += __uint_as_float(gr[i].x) * gv[i][0];
                    acc += __uint_as_float(gr[i].z) * gv[i][1];
                }
            }
            int tg = early ? NEA : 0;
            for (; tg + 4 <= T0; tg += 4) {          // eight corner rows in flight per lane
                u32x4 r[4];
                f32x4 v[4][2];
#pragma unroll
                for (int i = 0; i < 4; ++i) r[i] = rec_g[2 * (tg + i)];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i][0] = buf_load_ch4<TV>(vr, r[i].y + sub16);
                    v[i][1] = buf_load_ch4<TV>(vr, r[i].w + sub16);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc += __uint_as_float(r[i].x) * v[i][0];
                    acc += __uint_as_float(r[i].z) * v[i][1];
                }
            }
            for (; tg + 2 <= T0; tg += 2) {          // four
                u32x4 r[2];
                f32x4 v[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i) r[i] = rec_g[2 * (tg + i)];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    v[i][0] = buf_load_ch4<TV>(vr, r[i].y + sub16);
                    v[i][1] = buf_load_ch4<TV>(vr, r[i].w + sub16);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    acc += __uint_as_float(r[i].x) * v[i][0];
                    acc += __uint_as_float(r[i].z) * v[i][1];
                }
            }
            for (; tg < T0; ++tg) {
                const u32x4 r = rec_g[2 * tg];
                const f32x4 v0 = buf_load_ch4<TV>(vr, r.y + sub16), v1 = buf_load_ch4<TV>(vr, r.w + sub16);
                acc += __uint_as_float(r.x) * v0;
                acc += __uint_as_float(r.z) * v1;
            }
            finish_row(acc);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        WIN_STAMP();   // 9 + 3 it: gathered, stored
    }
    }
#undef WIN_STAMP
    if (pl.stats != nullptr) {      // kernel selection: this launch's counts out, the totals so far to the host
        if (stat_wg) {
            // the share's denominator: the windowed points of the rows this wavefront staged (counted once, here --
            // a second ballot per step in the staging loop cost 0.5 us per launch)
            unsigned n_rows = 0;
            for (int k0 = 0; k0 < iters; k0 += 16) {      // sixteen steps (x four rows) per ballot
                const int k = k0 + (lane >> 2), st = wave + k * nw;
                const bool mine = k < iters && st < steps && s_rowq[st * 4 + (lane & 3)] >= 0;
                n_rows += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(mine));
            }
            const unsigned n_live = n_rows * (unsigned)(LP - T0);
            unsigned n_off = v_off;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) n_off += __shfl_xor(n_off, o, 64);
            // (a launch whose windows were placed without measured means says nothing about the offsets)
            if (lane == 0 && hint_valid) sel_add(pl.stats, pl.sel_level, (unsigned)((sw >> 3) * nw + wave), n_live, n_off, 0u);
        }
        if (sw == 0 && wave == 1) {
            sel_publish(pl.stats, pl.stats_host, lane);
            if (!measure) sel_hint_publish(pl.stats, lane, M, L);
        }
    }
}

// Plan the region tiling from the HOST copy of the level shapes; false when this kernel does not apply.
// margins[l]: window side on level l = region side + 2 * margin + 1.
inline bool make_win_plan(WinPlan &pl, const int64_t *shapes_host, int N, int S, int M, int D, int L, int Lq, int P,
                          long value_bytes, int rlogx, int rlogy, int lwin0, const int *margins, int threads,
                          size_t &lds, int elem_bytes = 4) {
    const int gplog = elem_bytes == 4 ? 3 : 4, gp = 1 << gplog;      // pixels per 1-KiB fill group
    if (!shapes_host || D != 32 || L < 1 || L > kWinMaxL || Lq != S || L * P > 16) return false;
    if (rlogy < L - 1) rlogy = L - 1;
    if (rlogx < rlogy) rlogx = rlogy;
    if (rlogx > 5 || rlogy > 5 || threads < 64 || threads > 512 || (threads & 63)) return false;
    if (lwin0 < 0) lwin0 = 0;
    if (lwin0 > L) lwin0 = L;
    memset(&pl, 0, sizeof(pl));
    pl.N = N; pl.S = S; pl.M = M; pl.L = L; pl.P = P; pl.Lq = Lq; pl.lwin0 = lwin0; pl.rlogx = rlogx; pl.rlogy = rlogy;
    pl.rsx = 1 << rlogx; pl.rsy = 1 << rlogy; pl.grid = 0;
    pl.value_bytes = (unsigned)value_bytes;
    long q = 0;
    int rows = 0, px = 0, RY = 0, RX = 0;
    for (int l = 0; l < kWinMaxL; ++l) {
        if (l < L) {
            const long H = shapes_host[2 * l], W = shapes_host[2 * l + 1];
            if (H <= 0 || W <= 0 || H > 32767 || W > 32767 || W * M >= (1L << 23) || H * W >= (1L << 23)) return false;
            const int sx = rlogx - l, sy = rlogy - l, side_x = 1 << sx, side_y = 1 << sy;
            int ww = 0, wh = 0;
            if (l >= lwin0) {
                int mg = margins ? margins[l] : 3;
                if (mg < 0) mg = 0;
                ww = side_x + 2 * mg + 1;
                wh = side_y + 2 * mg + 1;
                if (ww > (int)W + 2) ww = (int)W + 2;      // never larger than the level and its zero border
                if (wh > (int)H + 2) wh = (int)H + 2;
                if (ww < 2) ww = 2;
                if (wh < 2) wh = 2;
            }
            pl.H[l] = (int)H; pl.W[l] = (int)W; pl.qstart[l] = (int)q; pl.shx[l] = sx; pl.shy[l] = sy; pl.row0[l] = rows;
            pl.rcpH[l] = (float)(1.0 / (double)H); pl.rcpW[l] = (float)(1.0 / (double)W);
            pl.ww[l] = ww; pl.wh[l] = wh; pl.wbase[l] = px;
            int magic = 65537;
            if (ww > 0) {
                if (ww * wh >= 32768) return false;
                magic = 65536 / ww + 1;
                for (int x = 0; x < ww * wh; ++x)
                    if (((x * magic) >> 16) != x / ww) return false;
            }
            pl.wmagic[l] = magic;
            q += H * W; rows += side_x * side_y; px += (ww * wh + gp - 1) & ~(gp - 1);     // windows start on group boundaries
            if ((ww * wh + gp - 1) / gp > pl.wgroups_max) pl.wgroups_max = (ww * wh + gp - 1) / gp;
            const int ry = (int)((H + side_y - 1) / side_y), rx = (int)((W + side_x - 1) / side_x);
            RY = ry > RY ? ry : RY;
            RX = rx > RX ? rx : RX;
        } else {
            pl.H[l] = 1; pl.W[l] = 1; pl.qstart[l] = (int)q; pl.shx[l] = 0; pl.shy[l] = 0; pl.row0[l] = rows;
            pl.rcpH[l] = pl.rcpW[l] = 1.f;
            pl.ww[l] = 0; pl.wh[l] = 0; pl.wmagic[l] = 65537; pl.wbase[l] = px;
        }
    }
    if (q != S) return false;      // the host shapes do not describe this value tensor
    // 24-bit multiplies in the step loop: (query, head) rows, projection-row offsets' factors and row strides in range
    if ((long)N * Lq * M >= (1L << 24) || (long)M * 32 * elem_bytes >= (1L << 23) || L * P > 16) return false;
    for (int l = L; l <= kWinMaxL; ++l) { pl.row0[l] = rows; pl.wbase[l] = px; }
    pl.rows = rows; pl.steps = (rows + 3) / 4; pl.RY = RY; pl.RX = RX;
    pl.RYf = (int)(shapes_host[0] >> rlogy); pl.RXf = (int)(shapes_host[1] >> rlogx);
    if (pl.RYf > RY) pl.RYf = RY;
    if (pl.RXf > RX) pl.RXf = RX;
    pl.rcpP = (float)(1.0 / (double)P);
    for (int a = 0; a < kWinMaxL; ++a)
        for (int c = 0; c < kWinMaxL; ++c) {
            pl.ratw[a][c] = (float)((double)pl.W[c] / (double)pl.W[a]);
            pl.rath[a][c] = (float)((double)pl.H[c] / (double)pl.H[a]);
        }
    pl.groups = lwin0 < L ? px / gp + 1 : 0;
    pl.gplog = gplog;
    const long nb = (long)N * RY * RX * M;
    if (nb > (1L << 30)) return false;
    pl.n_blocks = (int)nb;
    lds = (size_t)pl.groups * 1024 + (size_t)(threads / 64) * 2048 + (size_t)pl.steps * 16;
    return lds <= 160 * 1024 - 4096;
}


// Grid mode (round 6, late): the finest level is cut into ceil(H0 / rsy) x ceil(W0 / rsx) EQUAL regions of any size, and
// every coarser pixel goes to the region its centre falls into (win_bound).  Why: the power-of-two regions of an
// 800 x 1333 image are 77 per head -- 60 whole, 17 partial -- on the 64 workgroup slots of the XCD that head runs on; the
// partial ones start when a slot frees and pay a whole prologue each, a quarter of the launch at N = 1.  63 equal regions
// (12 x 24 pixels) are one round.  Same kernel: the per-region tables already hold a rectangle per level.
// Window of level l: ceil(rs H_l / H_0) pixels + 2 margin + 1 (the misalignment of a region against the coarser pixels is
// paid out of the margin; results never depend on the windows, only the speed does).
inline bool make_win_plan_grid(WinPlan &pl, const int64_t *shapes_host, int N, int S, int M, int D, int L, int Lq, int P,
                               long value_bytes, int rsy, int rsx, int lwin0, const int *margins, int threads,
                               size_t &lds, int elem_bytes = 4) {
    const int gplog = elem_bytes == 4 ? 3 : 4, gp = 1 << gplog;
    if (!shapes_host || D != 32 || L < 1 || L > kWinMaxL || Lq != S || L * P > 16) return false;
    if (rsy < 1 || rsx < 1 || rsy > 64 || rsx > 64 || threads < 64 || threads > 512 || (threads & 63)) return false;
    if (lwin0 < 0) lwin0 = 0;
    if (lwin0 > L) lwin0 = L;
    memset(&pl, 0, sizeof(pl));
    pl.N = N; pl.S = S; pl.M = M; pl.L = L; pl.P = P; pl.Lq = Lq; pl.lwin0 = lwin0;
    pl.rsx = rsx; pl.rsy = rsy; pl.grid = 1; pl.rlogx = pl.rlogy = 0;
    pl.value_bytes = (unsigned)value_bytes;
    const long H0 = shapes_host[0], W0 = shapes_host[1];
    if (H0 <= 0 || W0 <= 0) return false;
    const int RY = (int)((H0 + rsy - 1) / rsy), RX = (int)((W0 + rsx - 1) / rsx);
    long q = 0;
    int px = 0;
    for (int l = 0; l < kWinMaxL; ++l) {
        if (l < L) {
            const long H = shapes_host[2 * l], W = shapes_host[2 * l + 1];
            if (H <= 0 || W <= 0 || H > 32767 || W > 32767 || W * M >= (1L << 23) || H * W >= (1L << 23)) return false;
            if (H > H0 || W > W0) return false;            // (a pyramid: no level finer than the first)
            const int side_y = (int)((rsy * H + H0 - 1) / H0), side_x = (int)((rsx * W + W0 - 1) / W0);
            int ww = 0, wh = 0;
            if (l >= lwin0) {
                int mg = margins ? margins[l] : 3;
                if (mg < 0) mg = 0;
                ww = side_x + 2 * mg + 1;
                wh = side_y + 2 * mg + 1;
                if (ww > (int)W + 2) ww = (int)W + 2;
                if (wh > (int)H + 2) wh = (int)H + 2;
                if (ww < 2) ww = 2;
                if (wh < 2) wh = 2;
            }
            pl.H[l] = (int)H; pl.W[l] = (int)W; pl.qstart[l] = (int)q; pl.shx[l] = 0; pl.shy[l] = 0;
            pl.rcpH[l] = (float)(1.0 / (double)H); pl.rcpW[l] = (float)(1.0 / (double)W);
            pl.ww[l] = ww; pl.wh[l] = wh; pl.wbase[l] = px;
            int magic = 65537;
            if (ww > 0) {
                if (ww * wh >= 32768) return false;
                magic = 65536 / ww + 1;
                for (int x = 0; x < ww * wh; ++x)
                    if (((x * magic) >> 16) != x / ww) return false;
            }
            pl.wmagic[l] = magic;
            q += H * W; px += (ww * wh + gp - 1) & ~(gp - 1);
            if ((ww * wh + gp - 1) / gp > pl.wgroups_max) pl.wgroups_max = (ww * wh + gp - 1) / gp;
        } else {
            pl.H[l] = 1; pl.W[l] = 1; pl.qstart[l] = (int)q;
            pl.rcpH[l] = pl.rcpW[l] = 1.f;
            pl.ww[l] = 0; pl.wh[l] = 0; pl.wmagic[l] = 65537; pl.wbase[l] = px;
        }
    }
    if (q != S) return false;
    if ((long)N * Lq * M >= (1L << 24) || (long)M * 32 * elem_bytes >= (1L << 23)) return false;
    pl.wbase[kWinMaxL] = px;
    // rows of the largest region (the row table's size); a row of a region must stay below 33 pixels and a level of a
    // region below 1024 (win_row's division)
    int rows_max = 0;
    for (int ry = 0; ry < RY; ++ry)
        for (int rx = 0; rx < RX; ++rx) {
            int rows = 0;
            for (int l = 0; l < L; ++l) {
                const int hv = (ry + 1 == RY ? pl.H[l] : win_bound(ry + 1, rsy, pl.H[l], pl.H[0])) - win_bound(ry, rsy, pl.H[l], pl.H[0]);
                const int wv = (rx + 1 == RX ? pl.W[l] : win_bound(rx + 1, rsx, pl.W[l], pl.W[0])) - win_bound(rx, rsx, pl.W[l], pl.W[0]);
                if (hv < 0 || wv < 0 || wv > 32 || hv * wv >= 1024) return false;
                rows += hv * wv;
            }
            rows_max = rows > rows_max ? rows : rows_max;
        }
    if (rows_max < 1) return false;
    for (int l = 0; l <= kWinMaxL; ++l) pl.row0[l] = 0;      // (per region: the kernel's tables)
    pl.rows = rows_max; pl.steps = (rows_max + 3) / 4; pl.RY = RY; pl.RX = RX; pl.RYf = RY; pl.RXf = RX;
    pl.rcpP = (float)(1.0 / (double)P);
    for (int a = 0; a < kWinMaxL; ++a)
        for (int c = 0; c < kWinMaxL; ++c) {
            pl.ratw[a][c] = (float)((double)pl.W[c] / (double)pl.W[a]);
            pl.rath[a][c] = (float)((double)pl.H[c] / (double)pl.H[a]);
        }
    pl.groups = lwin0 < L ? px / gp + 1 : 0;
    pl.gplog = gplog;
    const long nb = (long)N * RY * RX * M;
    if (nb > (1L << 30)) return false;
    pl.n_blocks = (int)nb;
    lds = (size_t)pl.groups * 1024 + (size_t)(threads / 64) * 2048 + (size_t)pl.steps * 16;
    return lds <= 160 * 1024 - 4096;
}
