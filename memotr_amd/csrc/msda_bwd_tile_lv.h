// msda_bwd_tile_lv.h -- backward with fixed-point LDS windows, one pyramid level per workgroup (round 2; `bwd_variant`
// 10 and fused calls without a workspace; the default is msda_bwd_bins.h).  Included by msda_hip.hip inside its
// anonymous namespace.
#pragma once

// ---- backward, fixed-point window accumulation -----------------------------------------------------
// The LDS windows accumulate grad_value as 32-bit fixed point (on gfx950 ds_add_f32 retires ~0.33 lanes/clk/CU,
// integer LDS atomics 5-13, profiles/r01_ubench_*).  Two channels travel in one ds_add_u64: the low word carries
// channel 2k, the high word channel 2k+1 (the low word's sign is folded into the high word, so the pair sums
// exactly: total = sum_hi * 2^32 + sum_lo in 64-bit two's complement).
//
// Scaling (all powers of two, so every conversion is exact):
//   * per channel c : |grad_out[:, c]| <= 2^gexp[c] over the region's rows
//   * per level   l : attention weights of level l <= 2^aexp[l]
//   * K = min(30 - ceil(log2(rows * P)), 21): a window cell receives at most rows*P contributions (one per point
//     of its level), each bounded by 2^K after scaling, so 32-bit sums cannot overflow; the per-contribution quantum
//     relative to its channel/level bound is 2^-K (2^-21 for L = P = 4).
//   * float -> fixed: bits(fma(w, s, 1.5 * 2^23)) - 0x4B400000 (one rounding, nearest-even; |w s| < 2^22)
// Rows whose gradient is >= 7 bits below the region's bounds in every channel, and regions that contain a
// non-finite gradient or weight, bypass the windows: their contributions go out as ordinary float atomics, exactly
// like the reference (ms_deform_im2col_cuda.cuh:149-152), so a large outlier cannot flush its neighbours to zero and
// NaN / Inf propagate.  The flush converts back and adds into grad_value with float atomics like every other path.
//
// Records: 32 bytes per (row, point)
//   [0] global byte offset of corner (h0, w0) | window cells of corners 00,01 | cells 10,11 | flags
//       cells of dead / out-of-window / bypassed corners point at the row slot's dump row (never flushed), so the
//       scatter needs no branches; flags: bits 0-3 corner alive, bits 4-7 corner takes the float path,
//       bits 8.. = W_l * M (pixel-row stride in rows)
//   [1] lh, lw, attention weight, attention weight * 2^-aexp[l]
//   after a point is processed words 1-3 of [0] are recycled for its results (d/dx, d/dy, d/dattn).
typedef float __attribute__((may_alias)) f32_alias;     // the records are staged as u32x4 and read back as floats

__device__ __forceinline__ void lds_add_pair(unsigned char *p, float w, float s_lo, float s_hi) {
    const unsigned fa = __float_as_uint(fmaf(w, s_lo, 12582912.f));
    const unsigned fb = __float_as_uint(fmaf(w, s_hi, 12582912.f));
    const unsigned lo = fa - 0x4B400000u;
    const unsigned hi = fb - 0x4B400001u + ((fa >> 22) & 1u);      // bit 22 of the biased float: low word >= 0
    __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(p), ((unsigned long long)hi << 32) | lo,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ int bound_exponent(unsigned abs_bits) {
    // |x| < 2^e for the finite float with these abs bits; zero / tiny values -> -100 (their scale stays finite)
    const int e = (int)(abs_bits >> 23) - 126;
    return (abs_bits == 0u || e < -100) ? -100 : e;
}

template <typename TV>
__device__ __forceinline__ f32x4 load_ch4(const TV *p);
template <>
__device__ __forceinline__ f32x4 load_ch4<float>(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
template <>
__device__ __forceinline__ f32x4 load_ch4<bf16_t>(const bf16_t *p) {
    const u32x2 u = *reinterpret_cast<const u32x2 *>(p);
    return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                 __uint_as_float(u.y & 0xffff0000u)};
}

// ---- backward, fixed-point windows, one pyramid level per workgroup ---------------------------------------
// msda_bwd_d32_tile_q2 keeps the windows of all L levels of a (region, head) in one workgroup: ~70 KB of LDS, two
// workgroups (8 wavefronts) per CU, and every workgroup is one long chain of dependent global round trips (bounds
// pass, placement pass, three staging passes) -- with parts removed one by one (tools/bwd_ablate.py) 190 of its 320 us
// remain with no value loads, no scatter and no flush at all: it is latency-bound, not throughput-bound.
// Here a workgroup owns (batch, region, head, LEVEL): the P points of that level of the region's 85 queries.
//   * LDS: one window (25 KB for level 0, <= 13 KB for the others) + 32 x (2P+1) records: 5-8 workgroups per CU;
//   * every global input of the workgroup (grad_out rows, the level's locations / weights, or offsets + logits +
//     reference points) is loaded ONCE, up front, for all three 32-row passes and kept in registers: bounds,
//     window placement and staging all work from those registers -- one round trip instead of eleven;
//   * the level's constants (H, W, window) are wave-uniform.
// The fixed-point scheme (64-bit packed LDS atomics, per-channel x per-level power-of-two scales, float path for
// non-finite regions and for small lanes of wide regions) is that of msda_bwd_d32_tile_q2.
// Fused mode: the softmax Jacobian couples the levels (grad_logit_t = a_t (ga_t - sum_j a_j ga_j)), so this kernel
// leaves the raw d/d(attention) in the logit columns of grad_proj and msda_softmax_jacobian_kernel finishes them in
// place.
struct LevelPlanRow {
    bool ok;
    int q;
};

__device__ __forceinline__ LevelPlanRow level_tile_row(const TilePlan &pl, int r, int ry, int rx) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < kTileMaxL; ++i)
        if (i < pl.L && r >= pl.row0[i]) l = i;
    int row0 = pl.row0[0], sh = pl.shift[0], H = pl.H[0], W = pl.W[0], qs = pl.qstart[0];
#pragma unroll
    for (int i = 1; i < kTileMaxL; ++i)
        if (l == i) { row0 = pl.row0[i]; sh = pl.shift[i]; H = pl.H[i]; W = pl.W[i]; qs = pl.qstart[i]; }
    const int local = r - row0;
    const int py = (ry << sh) + (local >> sh), px = (rx << sh) + (local & ((1 << sh) - 1));
    LevelPlanRow o;
    o.ok = (r < pl.rows) && (py < H) && (px < W);
    o.q = o.ok ? qs + py * W + px : 0;
    return o;
}

template <typename TV>
__device__ __forceinline__ f32x2 load_ch2(const TV *p);
template <>
__device__ __forceinline__ f32x2 load_ch2<float>(const float *p) { return *reinterpret_cast<const f32x2 *>(p); }
template <>
__device__ __forceinline__ f32x2 load_ch2<bf16_t>(const bf16_t *p) {
    const unsigned u = *reinterpret_cast<const unsigned *>(p);
    return f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
}

#ifndef MSDA_LV_WGS
#define MSDA_LV_WGS 4      // workgroups per CU the register budget of tile_lv<2> is sized for
#endif
template <int PTS, typename TV, bool FUSED>
__global__ __launch_bounds__(kTileThreads, PTS <= 2 ? MSDA_LV_WGS : 2) void msda_bwd_d32_tile_lv(
    const TV *__restrict__ value, const int64_t *__restrict__ lstart, const PointSrc src,
    const TV *__restrict__ grad_out, float *__restrict__ grad_value, float *__restrict__ grad_loc,
    float *__restrict__ grad_attn, float *__restrict__ grad_proj, const TilePlan pl) {
    constexpr int D = 32;
    constexpr int NP = (kTileMaxRows + 31) / 32;                 // passes of 32 rows
    constexpr unsigned ROWB = 32u * (unsigned)sizeof(TV);
    __shared__ unsigned s_gbits[D];
    __shared__ unsigned s_abits;
    __shared__ float s_cscale[D], s_cinv[D];
    __shared__ float s_lscale, s_linv;
    __shared__ int s_nonfinite;
    __shared__ unsigned s_rowrange[2];
    __shared__ float s_sum[3];
    __shared__ int s_org[2];
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];

    // ---- block -> (batch, region, head, level); XCD-aware like tile_block_coords ----
    const int nb_pad = gridDim.x, chunk = nb_pad >> 3;
    const int sw = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (sw >= pl.n_blocks * pl.L) return;
    const int L = pl.L, P = pl.P, LP = L * P;
    const int l = sw % L;
    const int id = sw / L;
    const int m = id % pl.M;
    const int reg = (id / pl.M) % (pl.RY * pl.RX);
    const int b = id / (pl.M * pl.RY * pl.RX);
    const int ry = reg / pl.RX, rx = reg - ry * pl.RX;
    int H = pl.H[0], W = pl.W[0], win = pl.win[0], magic = pl.win_magic[0], shl = pl.shift[0];
#pragma unroll
    for (int i = 1; i < kTileMaxL; ++i)
        if (l == i) { H = pl.H[i]; W = pl.W[i]; win = pl.win[i]; magic = pl.win_magic[i]; shl = pl.shift[i]; }
    const int lstart_l = (int)lstart[l];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 3, sub = lane & 7;
    if (threadIdx.x < D) s_gbits[threadIdx.x] = 0u;
    if (threadIdx.x == 0) {
        s_abits = 0u;
        s_nonfinite = 0;
        s_rowrange[0] = 0x7f800000u;
        s_rowrange[1] = 0u;
        s_sum[0] = s_sum[1] = s_sum[2] = 0.f;
    }
    const int win_px = win * win;
    u32x4 *win_u4 = reinterpret_cast<u32x4 *>(s_dyn);
    for (int i = threadIdx.x; i < (win_px + 8) * 8; i += kTileThreads) win_u4[i] = u32x4{0u, 0u, 0u, 0u};
    const int rec_stride = 2 * P + 1;
    const unsigned dump_cell = (unsigned)(win_px + grp);
    u32x4 *rec = reinterpret_cast<u32x4 *>(s_dyn + (size_t)(win_px + 8) * 128) + (size_t)(wave * 8 + grp) * rec_stride;
    const __amdgpu_buffer_rsrc_t vr = make_rsrc(value, pl.value_bytes);
    const __amdgpu_buffer_rsrc_t gr = make_rsrc(grad_value, (unsigned)((size_t)pl.N * pl.S * pl.M * D * 4u));
    const int rot = grp & 1;
    const int cpair[2] = {2 * sub + 16 * rot, 2 * sub + 16 * (rot ^ 1)};
    const long mask_base = (long)b * pl.S + lstart_l;

    // ---- phase A: every global input of this workgroup, once ----
    bool ok[NP];
    int qq[NP];
    f32x4 g[NP];
    float gsr[NP][4];
    float px_[NP], py_[NP], pa[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const LevelPlanRow row = level_tile_row(pl, p * 32 + wave * 8 + grp, ry, rx);
        ok[p] = row.ok && p * 32 < pl.rows;
        qq[p] = row.q;
        const unsigned qrow = (unsigned)b * (unsigned)pl.Lq + (unsigned)row.q;   // 32-bit: check_dims' envelope
        const unsigned pm = qrow * (unsigned)pl.M + (unsigned)m;
        g[p] = f32x4{0.f, 0.f, 0.f, 0.f};
        gsr[p][0] = gsr[p][1] = gsr[p][2] = gsr[p][3] = 0.f;
        px_[p] = py_[p] = pa[p] = 0.f;
        if (ok[p]) {
            g[p] = load_ch4<TV>(grad_out + (pm * (unsigned)D + (unsigned)(sub * 4)));
            const f32x2 s0 = load_ch2<TV>(grad_out + (pm * (unsigned)D + (unsigned)cpair[0])), s1 = load_ch2<TV>(grad_out + (pm * (unsigned)D + (unsigned)cpair[1]));
            gsr[p][0] = s0.x; gsr[p][1] = s0.y; gsr[p][2] = s1.x; gsr[p][3] = s1.y;
        }
        float mx = 0.f, rsum = 1.f;
        const float *lg = nullptr;
        if (FUSED) {     // (all 64 lanes: the row reductions are DPP)
            lg = fused_logits(src, qrow, m, LP);
            row_softmax_stats<8>(lg, LP, sub, mx, rsum);
        }
        if (ok[p] && sub < P) {
            const int t = l * P + sub;
            const f32x2 xy = point_location<FUSED>(src, pm, qrow, m, L, P, t, l, H, W);
            px_[p] = xy.x;
            py_[p] = xy.y;
            pa[p] = FUSED ? sm_exp(lg[t], mx) * rsum : src.attn[pm * (unsigned)LP + (unsigned)t];
        }
    }
    __syncthreads();      // the zeroed window / shared scalars are in place

    // ---- phase B: bounds and window placement, from registers ----
    {
        unsigned g4[4] = {0u, 0u, 0u, 0u}, rmin = 0x7f800000u, rmax = 0u, amax = 0u;
        float sx = 0.f, sy = 0.f, cnt = 0.f;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            unsigned rowm = 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned u = __float_as_uint(g[p][j]) & 0x7fffffffu;
                g4[j] = u > g4[j] ? u : g4[j];
                rowm = u > rowm ? u : rowm;
            }
            rowm = __float_as_uint(row_max<8>(__uint_as_float(rowm < 0x7f800000u ? rowm : 0x7f7fffffu)));
            if (ok[p] && rowm != 0u) {
                rmin = rowm < rmin ? rowm : rmin;
                rmax = rowm > rmax ? rowm : rmax;
            }
            if (ok[p] && sub < P) {
                const unsigned u = __float_as_uint(pa[p]) & 0x7fffffffu;
                amax = u > amax ? u : amax;
                // (w_low + lw, h_low + lh) of sample_setup = the un-floored pixel position; same gate
                const float w_im = px_[p] * (float)W - 0.5f, h_im = py_[p] * (float)H - 0.5f;
                if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
                    sx += w_im;
                    sy += h_im;
                    cnt += 1.f;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicMax(&s_gbits[sub * 4 + j], g4[j]);
        if (amax) atomicMax(&s_abits, amax);
        if (sub == 0 && rmax != 0u) {
            atomicMin(&s_rowrange[0], rmin);
            atomicMax(&s_rowrange[1], rmax);
        }
        sx = wave_sum(sx);
        sy = wave_sum(sy);
        cnt = wave_sum(cnt);
        if (lane == 0) {
            atomicAdd(&s_sum[0], sx);
            atomicAdd(&s_sum[1], sy);
            atomicAdd(&s_sum[2], cnt);
        }
    }
    __syncthreads();
    int cnt_log2 = 0;
    while ((1 << cnt_log2) < pl.rows * P) ++cnt_log2;
    int K = 30 - cnt_log2;
    K = K > 21 ? 21 : (K < 0 ? 0 : K);
    if (threadIdx.x <= D) {
        const unsigned bits = threadIdx.x < D ? s_gbits[threadIdx.x] : s_abits;
        if (bits >= 0x7e800000u) atomicOr(&s_nonfinite, 1);
    }
    if (threadIdx.x == 64) {      // window origin: centred on the mean sampling position of the level
        const float cnt = s_sum[2];
        const float cx = cnt > 0.f ? s_sum[0] / cnt : (float)((rx << shl) + (1 << shl) / 2);
        const float cy = cnt > 0.f ? s_sum[1] / cnt : (float)((ry << shl) + (1 << shl) / 2);
        int ox = (int)floorf(cx - 0.5f * (float)(win - 1) + 0.5f);
        int oy = (int)floorf(cy - 0.5f * (float)(win - 1) + 0.5f);
        const int max_x = W - win, max_y = H - win;
        ox = ox > max_x ? max_x : ox;
        oy = oy > max_y ? max_y : oy;
        s_org[0] = oy < 0 ? 0 : oy;
        s_org[1] = ox < 0 ? 0 : ox;
    }
    __syncthreads();
    const bool nonfinite = s_nonfinite != 0;
    if (nonfinite) K = 0;
    if (threadIdx.x < D) {
        const int e = nonfinite ? 0 : bound_exponent(s_gbits[threadIdx.x]);
        s_cscale[threadIdx.x] = ldexpf(1.f, K - e);
        s_cinv[threadIdx.x] = ldexpf(1.f, e - K);
    } else if (threadIdx.x == D) {
        const int e = nonfinite ? 0 : bound_exponent(s_abits);
        s_lscale = ldexpf(1.f, -e);
        s_linv = ldexpf(1.f, e);
    }
    __syncthreads();
    const int oy = s_org[0], ox = s_org[1];
    const float lscale = s_lscale, linv = s_linv;
    const bool wide = pl.wide_log2 > 0 && s_rowrange[1] != 0u &&
                      (int)(s_rowrange[1] >> 23) - (int)(s_rowrange[0] >> 23) >= pl.wide_log2;
    const float lane_limit = wide ? ldexpf(1.f, K - 7) : 0.f;
    float cs[4], ci[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        cs[j] = s_cscale[cpair[j >> 1] + (j & 1)];
        ci[j] = s_cinv[cpair[j >> 1] + (j & 1)];
    }
    const unsigned lane_off = (unsigned)sub * (ROWB / 8u);
    const unsigned ps = (unsigned)pl.M * ROWB, wps = (unsigned)W * ps;
    const unsigned gps = (unsigned)pl.M * 128u, gwps = (unsigned)W * gps;

    // ---- phase C: the three passes, no global loads left except the value corners ----
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (p * 32 >= pl.rows) break;
        const unsigned qrow = (unsigned)b * (unsigned)pl.Lq + (unsigned)qq[p];
        const unsigned pm = qrow * (unsigned)pl.M + (unsigned)m;
        float gs[4];
        bool lane_bypass = nonfinite;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            gs[j] = gsr[p][j] * cs[j];
            lane_bypass = lane_bypass || (fabsf(gs[j]) < lane_limit && gs[j] != 0.f);
        }
        const unsigned dump_or = lane_bypass ? 0xffffffffu : 0u;
        if (sub < P) {
            Sample<float> s = sample_setup<float>(px_[p], py_[p], H, W);
            const float a = (s.gate && ok[p]) ? pa[p] : 0.f;
            if (!s.gate) s.lh = s.lw = 0.f;
            const int h0 = s.h_low, w0 = s.w_low;
            const bool on = s.gate && ok[p];
            const bool okh0 = on && h0 >= 0, okh1 = on && h0 + 1 <= H - 1;
            const bool okw0 = w0 >= 0, okw1 = w0 + 1 <= W - 1;
            bool v00 = okh0 && okw0, v01 = okh0 && okw1, v10 = okh1 && okw0, v11 = okh1 && okw1;
            if (src.mask != nullptr) {      // (the split fused backward runs the plain instantiation with a mask)
                const unsigned char *mk = src.mask + mask_base;
                const int p00 = h0 * W + w0;
                v00 = v00 && !mk[v00 ? p00 : 0];
                v01 = v01 && !mk[v01 ? p00 + 1 : 0];
                v10 = v10 && !mk[v10 ? p00 + W : 0];
                v11 = v11 && !mk[v11 ? p00 + W + 1 : 0];
            }
            const int wy0 = h0 - oy, wx0 = w0 - ox;
            const bool iy0 = (unsigned)wy0 < (unsigned)win, iy1 = (unsigned)(wy0 + 1) < (unsigned)win;
            const bool ix0 = (unsigned)wx0 < (unsigned)win, ix1 = (unsigned)(wx0 + 1) < (unsigned)win;
            const unsigned c00 = (unsigned)(wy0 * win + wx0);
            const bool w00 = v00 && iy0 && ix0, w01 = v01 && iy0 && ix1, w10 = v10 && iy1 && ix0, w11 = v11 && iy1 && ix1;
            u32x4 r0v;
            r0v.x = (((unsigned)b * (unsigned)pl.S + (unsigned)(lstart_l + h0 * W + w0)) * (unsigned)pl.M + (unsigned)m) * ROWB;
            r0v.y = (w00 ? c00 : dump_cell) | ((w01 ? c00 + 1u : dump_cell) << 16);
            r0v.z = (w10 ? c00 + (unsigned)win : dump_cell) | ((w11 ? c00 + (unsigned)win + 1u : dump_cell) << 16);
            r0v.w = (unsigned)v00 | ((unsigned)v01 << 1) | ((unsigned)v10 << 2) | ((unsigned)v11 << 3) |
                    ((unsigned)(v00 && !w00) << 4) | ((unsigned)(v01 && !w01) << 5) |
                    ((unsigned)(v10 && !w10) << 6) | ((unsigned)(v11 && !w11) << 7);
            f32x4 w;
            w.x = s.lh;
            w.y = s.lw;
            w.z = a;
            w.w = a * lscale;
            rec[2 * sub] = r0v;
            rec[2 * sub + 1] = __builtin_bit_cast(u32x4, w);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const f32x4 gg = g[p];
        for (int t0 = 0; t0 < P; t0 += PTS) {
            u32x4 ra[PTS];
            f32x4 rw[PTS], v[PTS][4];
#pragma unroll
            for (int i = 0; i < PTS; ++i) {
                const int t = (t0 + i < P) ? t0 + i : P - 1;
                ra[i] = rec[2 * t];
                rw[i] = __builtin_bit_cast(f32x4, rec[2 * t + 1]);
            }
#pragma unroll
            for (int i = 0; i < PTS; ++i) {
                const unsigned fl = ra[i].w, base = ra[i].x + lane_off;
                const bool ld = !(pl.ablate & 4);
                v[i][0] = buf_load_ch4<TV>(vr, ((fl & 1u) && ld) ? base : kOobOffset);
                v[i][1] = buf_load_ch4<TV>(vr, ((fl & 2u) && ld) ? base + ps : kOobOffset);
                v[i][2] = buf_load_ch4<TV>(vr, ((fl & 4u) && ld) ? base + wps : kOobOffset);
                v[i][3] = buf_load_ch4<TV>(vr, ((fl & 8u) && ld) ? base + wps + ps : kOobOffset);
            }
#pragma unroll
            for (int i = 0; i < PTS; ++i) {
                const int t = t0 + i;
                if (t < P) {
                    const unsigned fl = ra[i].w;
                    const float lh = rw[i].x, lw = rw[i].y, a = rw[i].z, a_s = rw[i].w;
                    const float hh = 1.f - lh, hw = 1.f - lw;
                    const float wk[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
                    unsigned cells[4] = {ra[i].y & 0xffffu, ra[i].y >> 16, ra[i].z & 0xffffu, ra[i].z >> 16};
                    unsigned fpath = fl >> 4;
                    if (wide || nonfinite) {      // block-uniform: ordinary regions skip the per-lane overrides
#pragma unroll
                        for (int k = 0; k < 4; ++k) cells[k] = dump_or ? dump_cell : cells[k];
                        fpath |= dump_or & fl;
                    }
                    if (!(pl.ablate & 2))
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        unsigned char *pw_ = s_dyn + (cells[k] << 7);
                        const float wa = wk[k] * a_s;
                        lds_add_pair(pw_ + cpair[0] * 4, wa, gs[0], gs[1]);
                        lds_add_pair(pw_ + cpair[1] * 4, wa, gs[2], gs[3]);
                    }
                    if (!(pl.ablate & 2) && __builtin_amdgcn_ballot_w64((fpath & 0xfu) != 0u) != 0ull) {   // rare: float path
                        const unsigned gbase = ra[i].x * (128u / ROWB);
                        const unsigned dg[4] = {0u, gps, gwps, gwps + gps};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if (fpath & (1u << k)) {
                                const float wa = wk[k] * a_s;
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(
                                        (wa * gs[j]) * ci[j] * linv, gr,
                                        (int)(gbase + dg[k] + (unsigned)(cpair[j >> 1] + (j & 1)) * 4u), 0, 0);
                            }
                        }
                    }
                    const f32x4 tga = gg * a;
                    const f32x4 val = wk[0] * v[i][0] + wk[1] * v[i][1] + wk[2] * v[i][2] + wk[3] * v[i][3];
                    const f32x4 gw = hh * (v[i][1] - v[i][0]) + lh * (v[i][3] - v[i][2]);
                    const f32x4 gh = hw * (v[i][2] - v[i][0]) + lw * (v[i][3] - v[i][1]);
                    float ra_ = gg.x * val.x + gg.y * val.y + gg.z * val.z + gg.w * val.w;
                    float rw_ = gw.x * tga.x + gw.y * tga.y + gw.z * tga.z + gw.w * tga.w;
                    float rh_ = gh.x * tga.x + gh.y * tga.y + gh.z * tga.z + gh.w * tga.w;
                    ra_ = sum8(ra_);
                    rw_ = sum8(rw_);
                    rh_ = sum8(rh_);
                    if (sub == t) {
                        f32_alias *slot = reinterpret_cast<f32_alias *>(&rec[2 * t]);
                        slot[1] = rw_;
                        slot[2] = rh_;
                        slot[3] = ra_;
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (ok[p]) {
            const f32_alias *res = reinterpret_cast<const f32_alias *>(rec);
            for (int i = sub; i < 2 * P; i += 8) {
                const int t = i >> 1, comp = i & 1;
                const float size = (float)(comp ? H : W);
                const float r_ = res[8 * t + 1 + comp];       // d/d(pixel position); grad_loc = r_ * size
                if (FUSED) {
                    float go = r_;                            // 2-d: (r_ * size) / size
                    if (src.ref_dim != 2) {
                        const float *rp = src.ref + (qrow * (unsigned)L + (unsigned)l) * 4u;
                        go = (r_ * size) * (rp[2 + comp] * (0.5f / (float)P));
                    }
                    grad_proj[qrow * (unsigned)src.proj_stride + (unsigned)(m * 2 * LP + l * P * 2 + i)] = go;
                } else if (grad_proj != nullptr) {            // split fused backward: d/d loc parked in the offset columns
                    grad_proj[qrow * (unsigned)src.proj_stride + (unsigned)(m * 2 * LP + l * P * 2 + i)] = r_ * size;
                } else {
                    grad_loc[pm * (unsigned)(LP * 2) + (unsigned)(l * P * 2 + i)] = r_ * size;
                }
            }
            if (sub < P) {
                if (FUSED || grad_proj != nullptr)
                    grad_proj[qrow * (unsigned)src.proj_stride + (unsigned)(src.n_off + m * LP + l * P + sub)] = res[8 * sub + 3];
                else
                    grad_attn[pm * (unsigned)LP + (unsigned)(l * P + sub)] = res[8 * sub + 3];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // ---- flush: one coalesced global float atomic per touched window element ----
    if (!(pl.ablate & 1)) {
        const unsigned long long *win_u64 = reinterpret_cast<const unsigned long long *>(s_dyn);
        const int c = threadIdx.x & 31;                       // this thread's channel in every pixel row it visits
        const bool high = (c & 1) != 0;
        const float back = s_cinv[c] * linv;                  // powers of two: exact
        const unsigned col = (unsigned)m * 128u + (unsigned)c * 4u;
        for (int pix = threadIdx.x >> 5; pix < win_px; pix += kTileThreads / 32) {
            const unsigned long long tot = win_u64[pix * (D / 2) + (c >> 1)];
            const int lo = (int)(unsigned)(tot & 0xffffffffull);
            const int q = high ? (int)(unsigned)(tot >> 32) + (lo < 0 ? 1 : 0) : lo;
            if (q != 0) {
                const int wy = (pix * magic) >> 16, wx = pix - wy * win;
                const int gy = oy + wy, gx = ox + wx;
                if (gy < H && gx < W)
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(
                        (float)q * back, gr,
                        (int)(((unsigned)b * (unsigned)pl.S + (unsigned)(lstart_l + gy * W + gx)) * (unsigned)pl.M * 128u +
                              col), 0, 0);
            }
        }
    }
}
