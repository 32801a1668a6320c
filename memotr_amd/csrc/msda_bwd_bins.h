// msda_bwd_bins.h -- region-tiled backward for pyramid self-attention (Lq == S, D = 32), round 4.
//
// Reference semantics: ms_deformable_col2im_gpu_kernel_* + ms_deform_attn_col2im_bilinear
// (models/ops/src/cuda/ms_deform_im2col_cuda.cuh:87-159, 301-403): per (query, head, level, point)
//   grad_value[corner k] += w_k * attn * grad_out_row          (4 corners x 32 channels, atomics)
//   grad_attn            = <grad_out_row, sum_k w_k v_k>
//   grad_loc             = attn * <grad_out_row, d(sum_k w_k v_k)/d(x, y)> * (W, H)
//
// msda_bwd_d32_tile_lv (round 2) did the grad_value part as a SCATTER: every (row, point, corner) converts its 32
// products to fixed point and adds them to an LDS window with packed 64-bit integer atomics (ds_add_f32 retires 15x
// slower) -- 106 M VALU instructions per encoder call, 74 % VALU-busy, 215 us.  This kernel turns the scatter into a
// GATHER, which needs neither fixed point nor per-element atomics:
//
//   * a workgroup owns (batch, region, head, level) like tile_lv: the 8x8 + 4x4 + 2x2 + 1 = 85 query rows of one
//     region x the P points of one level = 340 ITEMS, one lane each for all scalar work (sample arithmetic, masks,
//     window cells, bilinear weights) -- done once, not by the 8 lanes of a row redundantly;
//   * the window of the level is only a table of COUNTERS (4 B per cell instead of 128 B), so it can be generous;
//     every valid corner takes a ticket in its cell (one returning ds_add_u32 per corner instead of 16 ds_add_u64),
//     a prefix sum over the cells turns tickets into positions: a counting sort of the <= 1360 (cell, row, weight)
//     entries by cell, entirely in LDS;
//   * grad_out rows of the region are staged once in LDS (fp32); four lanes x 8 channels own a cell and walk its
//     entry list: acc += weight * grad_out_row -- float accumulation in registers, 2 pk_fma per 4 channels, no
//     conversions, no scales, no "wide region" or non-finite bypass: NaN / Inf propagate like the reference's
//     atomicAdd does;
//   * the three gradient dot products of a point all derive from the FOUR corner dot products
//     d_k = <grad_out_row, v_k>: grad_attn = sum_k w_k d_k, d/dx = a (hh (d1 - d0) + lh (d3 - d2)),
//     d/dy = a (hw (d2 - d0) + lw (d3 - d1)); four lanes x 8 channels per item, reduced with two DPP steps;
//   * a cell's 32 sums leave as ONE 128-byte row of float atomics (32 consecutive lanes, the L2 atomic units' fastest
//     pattern), after a per-wavefront transpose through LDS.
// Points whose corners fall outside the window take global float atomics on the spot (results never depend on the
// window placement, only the speed does).
#pragma once

constexpr unsigned kBinsGRow = 144u;          // LDS bytes per staged grad_out row: 128 + 16 (bank spread for b128 reads)
constexpr unsigned kBinsStageWave = 1024u;    // flush transpose: 8 cells x 128 B per wavefront and pass

#ifndef MSDA_BINS_WGS
#define MSDA_BINS_WGS 5       // workgroups per CU the register budget is sized for (LDS: ~31 KB per workgroup)
#endif

struct BinsPlan {
    int n_items;             // rows * P
    int magic_p;             // (i * magic_p) >> 16 == i / P for i < 256 * NI
    int scan_c;              // cells per lane in the prefix sum (multiple of 4); counters are padded to 64 * scan_c
    int strip;               // region rows per strip of the block -> region walk (1: raster order)
    int shrink, level;           // statistics: also count the corners outside the window shrunk by `shrink` pixels; selector level
    unsigned long long *stats, *stats_host;   // msda_select.h records (device / mapped host), null: no statistics
    // byte offsets into dynamic LDS (grad_out rows at 0).  Two unions: o_x holds the items' records (16 B each) and
    // flags (at o_fl) until the row phase is over, the sorted entries afterwards; o_st holds the ticket counters and the
    // row tables until the sort is done, the flush transpose afterwards.  o_start: u16 per cell, o_comp: u32 per
    // non-empty cell (cell | count << 16), o_misc: 4 sums + 64 flush offsets + 3 counters.
    unsigned o_x, o_fl, o_st, o_rowp, o_rowa, o_rowq, o_cnt, o_start, o_comp, o_misc;
    int fused_loc;           // split fused backward: locations from the raw projection + reference points (src.proj / src.ref)
    int offsets_done;        // ... and final offset gradients (2-d reference points: d loc / d offset = 1 / (W, H))
    // soft (round 6): no side kernel at all.  The item's lane forms its softmax weight from the row's sixteen logits
    // (msda_common.h's arithmetic, the bits every other kernel forms) and the row phase writes the FINAL logit gradient
    // a_t (ga_t - sum_j a_j ga_j): the sum over the row's points is <grad_out_row, out_row> -- out = sum_j a_j sampled_j
    // is the forward's output, which the caller still holds -- so no level needs another level's results
    int soft;
    unsigned o_dot;          // soft: <grad_out_row, out_row> per staged row
};

__device__ __forceinline__ unsigned bins_incl_scan(unsigned x, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    return x;
}

// sum over the wavefront, DPP inside the rows of 16 lanes, the four row sums through SGPRs; every lane gets the sum
__device__ __forceinline__ float bins_wave_sum(float x) {
    x += MSDA_DPP(x, 0xB1);      // quad_perm [1,0,3,2]
    x += MSDA_DPP(x, 0x4E);      // quad_perm [2,3,0,1]
    x += MSDA_DPP(x, 0x141);     // row_half_mirror
    x += MSDA_DPP(x, 0x140);     // row_mirror
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 48));
    return (r0 + r1) + (r2 + r3);
}

// four consecutive channels of a grad_out row as floats
template <typename TV>
__device__ __forceinline__ f32x4 bins_load_g4(const TV *p);
template <>
__device__ __forceinline__ f32x4 bins_load_g4<float>(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
template <>
__device__ __forceinline__ f32x4 bins_load_g4<bf16_t>(const bf16_t *p) {
    const u32x2 u = *reinterpret_cast<const u32x2 *>(p);
    return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                 __uint_as_float(u.y & 0xffff0000u)};
}

// <ga, va> + <gb, vb> over 8 channels as packed pairs (v_pk_fma_f32 on the natural register pairs)
__device__ __forceinline__ float bins_dot8(const f32x4 ga, const f32x4 va, const f32x4 gb, const f32x4 vb) {
    f32x2 s = f32x2{ga.x, ga.y} * f32x2{va.x, va.y};
    s += f32x2{ga.z, ga.w} * f32x2{va.z, va.w};
    s += f32x2{gb.x, gb.y} * f32x2{vb.x, vb.y};
    s += f32x2{gb.z, gb.w} * f32x2{vb.z, vb.w};
    return s.x + s.y;
}

__device__ __forceinline__ int bins_mul24(int a, int b) { return __mul24(a, b); }

#define BINS_DPP_U(x, ctrl) ((unsigned)__builtin_amdgcn_update_dpp(0, (int)(x), (ctrl), 0xF, 0xF, true))

#define MSDA_QUAD_SUM(x)                                          \
    do {                                                          \
        (x) += MSDA_DPP((x), 0xB1); /* quad_perm [1,0,3,2] */     \
        (x) += MSDA_DPP((x), 0x4E); /* quad_perm [2,3,0,1] */     \
    } while (0)

template <int NI, typename TV, bool SOFT = false>
__global__ __launch_bounds__(kTileThreads, MSDA_BINS_WGS) void msda_bwd_d32_bins(
    const TV *__restrict__ value, const int64_t *__restrict__ lstart, const PointSrc src,
    const TV *__restrict__ grad_out, float *__restrict__ grad_value, float *__restrict__ grad_loc,
    float *__restrict__ grad_attn, float *__restrict__ grad_proj, const TilePlan pl, const BinsPlan bp,
    const TV *__restrict__ fwd_out = nullptr) {
    constexpr int D = 32;
    constexpr unsigned ROWB = 32u * (unsigned)sizeof(TV);
    constexpr bool kB16 = sizeof(TV) == 2;
    constexpr unsigned kChunkDelta = kB16 ? 16u : 64u;      // between a lane's two 16-byte chunks of a staged grad_out row
    // (no static LDS: the dynamic block then starts at LDS address 0 and every offset below folds into the instructions)
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];

    // ---- block -> (batch, region, head, level); XCD-aware like tile_lv ----
    const int nb_pad = gridDim.x, chunk = nb_pad >> 3;
    const int sw = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (sw >= pl.n_blocks * pl.L) return;
    const int L = pl.L, P = pl.P, LP = L * P, M = pl.M;
    const int l = sw % L;
    const int id = sw / L;
    const int m = id % M;
    const int reg = (id / M) % (pl.RY * pl.RX);
    const int b = id / (M * pl.RY * pl.RX);
    // Regions are walked in strips of `strip` region rows, column by column inside a strip: the windows of vertical
    // neighbours overlap as much as those of horizontal ones, and in raster order a vertical neighbour comes a whole
    // region row (thousands of workgroups, tens of MB through this XCD's L2) later -- its flush then finds the shared
    // grad_value lines evicted.  In strip order both kinds of neighbour are a few dozen workgroups apart.
    const int sh_rows = bp.strip;
    const int strip = reg / (sh_rows * pl.RX);
    const int rem = reg - strip * sh_rows * pl.RX;
    const int hgt = pl.RY - sh_rows * strip < sh_rows ? pl.RY - sh_rows * strip : sh_rows;
    const int rx = rem / hgt, ry = sh_rows * strip + (rem - rx * hgt);
    int H = pl.H[0], W = pl.W[0], win = pl.win[0], magic = pl.win_magic[0], shl = pl.shift[0], lstart_l = pl.qstart[0];
#pragma unroll
    for (int i = 1; i < kTileMaxL; ++i)
        if (l == i) { H = pl.H[i]; W = pl.W[i]; win = pl.win[i]; magic = pl.win_magic[i]; shl = pl.shift[i]; lstart_l = pl.qstart[i]; }
    // (one query per pixel: the level starts where its queries do; the device copy of level_start_index -- a cold
    //  scalar load in front of everything else -- is not read)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rows = pl.rows, n_items = bp.n_items;

    unsigned char *const G = s_dyn;
    u32x4 *const R = reinterpret_cast<u32x4 *>(s_dyn + bp.o_x);
    unsigned *const FL = reinterpret_cast<unsigned *>(s_dyn + bp.o_fl);
    unsigned *const ROWP = reinterpret_cast<unsigned *>(s_dyn + bp.o_rowp);
    unsigned *const ROWA = reinterpret_cast<unsigned *>(s_dyn + bp.o_rowa);
    unsigned *const ROWQ = reinterpret_cast<unsigned *>(s_dyn + bp.o_rowq);
    unsigned *const CNT = reinterpret_cast<unsigned *>(s_dyn + bp.o_cnt);
    unsigned short *const START = reinterpret_cast<unsigned short *>(s_dyn + bp.o_start);
    unsigned *const COMP = reinterpret_cast<unsigned *>(s_dyn + bp.o_comp);
    float *const s_sum = reinterpret_cast<float *>(s_dyn + bp.o_misc);
    unsigned *const s_off = reinterpret_cast<unsigned *>(s_dyn + bp.o_misc + 16u) + wave * 16;   // flush: grad_value byte offsets of this wavefront's 16 cells
    const __amdgpu_buffer_rsrc_t vr = make_rsrc(value, pl.value_bytes);
    const __amdgpu_buffer_rsrc_t gr = make_rsrc(grad_value, (unsigned)((size_t)pl.N * pl.S * M * D * 4u));

    // ---- phase -1: the region's row table (one level_tile_row per row, not per use), zeroed counters ----
    if (tid < 4) s_sum[tid] = 0.f;
    for (int i = tid; i < bp.scan_c * 64; i += kTileThreads) CNT[i] = 0u;
    unsigned *const s_cnt = reinterpret_cast<unsigned *>(s_dyn + bp.o_misc + 16u + 256u);   // valid / off / inner corners
    // statistics (msda_select.h): one workgroup in eight counts -- a sample of > 1000 workgroups per launch
    const bool stat_wg = bp.stats != nullptr && (sw & 7) == 0;
    if (tid < 3) s_cnt[tid] = 0u;
    if (tid <= rows) {                                  // (tid == rows: the zero row, not ok)
        const LevelPlanRow row = level_tile_row(pl, tid, ry, rx);
        const unsigned qrow = (unsigned)b * (unsigned)pl.Lq + (unsigned)row.q;
        const unsigned pm = qrow * (unsigned)M + (unsigned)m;
        ROWP[tid] = row.ok ? pm : 0xffffffffu;
        // first output element of the row (floats): a row of grad_proj (split fused backward) or of grad_attn (plain;
        // grad_loc is twice that)
        ROWA[tid] = grad_proj != nullptr ? qrow * (unsigned)src.proj_stride : pm * (unsigned)LP;
        ROWQ[tid] = qrow;
    }
    __syncthreads();      // B0

    // ---- phase 0: grad_out rows -> LDS, this thread's items.  Every global input of the phase is requested before
    //      the first one is used (round 5; the staging loop and the items used to be a chain of seven dependent round
    //      trips per workgroup: three staging iterations, then location -> attention weight per item).  Rows that do
    //      not exist read row 0 and are zeroed afterwards: no branch stands between the requests ----
    constexpr int kStageIters = ((kTileMaxRows + 1) * 8 + kTileThreads - 1) / kTileThreads;
    const int n_stage = (rows + 1) * 8;
    f32x4 st_g[kStageIters], st_o[kStageIters];
    bool st_ok[kStageIters];
#pragma unroll
    for (int i = 0; i < kStageIters; ++i) {
        const int idx = tid + i * kTileThreads;
        const int r = (idx >> 3) < rows ? (idx >> 3) : rows;        // (row `rows` is the zero row)
        const unsigned pm = ROWP[r];
        st_ok[i] = pm != 0xffffffffu;
        st_g[i] = bins_load_g4<TV>(grad_out + ((st_ok[i] ? pm : 0u) * (unsigned)D + (unsigned)((idx & 7) * 4)));
        st_o[i] = SOFT ? bins_load_g4<TV>(fwd_out + ((st_ok[i] ? pm : 0u) * (unsigned)D + (unsigned)((idx & 7) * 4)))
                          : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    int it_r[NI];
    bool it_live[NI], it_gate[NI];
    int it_h0[NI], it_w0[NI];
    float it_lh[NI], it_lw[NI], it_a[NI], it_araw[NI];
    {
        f32x2 raw[NI], r01[NI], r23[NI];
        float araw[NI];
        f32x4 lg4[NI];
        bool okk[NI];
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int it = tid + k * kTileThreads;
            const bool live = it < n_items;
            const int itc = live ? it : 0;
            const int r = bins_mul24(itc, bp.magic_p) >> 16, p = itc - bins_mul24(r, P);
            it_r[k] = r;
            const unsigned pm = ROWP[r];
            okk[k] = live && pm != 0xffffffffu;
            const unsigned pmc = okk[k] ? pm : 0u;
            const unsigned t = (unsigned)(l * P + p);
            const unsigned qrow = ROWQ[r];           // (a real query row even when the row does not exist)
            // (fused_loc: the module's own arithmetic, ms_deform_attn.py:114-120, the bits msda_fused_points_f32
            //  exposes.  The source is chosen by address, not by branch -- a branch between two requests makes the
            //  second wait for the first; operands a mode does not have re-read the first address)
            const float *const p_raw = bp.fused_loc ? src.proj + (qrow * (unsigned)src.proj_stride + (unsigned)((m * LP) * 2) + t * 2u)
                                                    : src.loc + (pmc * (unsigned)LP + t) * 2u;
            const float *const rp = src.ref + (qrow * (unsigned)L + (unsigned)l) * (unsigned)src.ref_dim;
            raw[k] = *reinterpret_cast<const f32x2 *>(p_raw);
            r01[k] = *reinterpret_cast<const f32x2 *>(bp.fused_loc ? rp : p_raw);
            r23[k] = *reinterpret_cast<const f32x2 *>(bp.fused_loc && src.ref_dim != 2 ? rp + 2 : p_raw);
            if (SOFT) {          // (L * P = 16: the row's logits.  P = 4: the row's four items of this level are an aligned
                                 //  quad of lanes -- each takes a quarter of the logits, the softmax is shared by DPP)
                const f32x4 *lp = reinterpret_cast<const f32x4 *>(src.proj + (qrow * (unsigned)src.proj_stride + (unsigned)src.n_off + (unsigned)(m * 16)));
                lg4[k] = lp[p];
                araw[k] = 0.f;
            } else {
                araw[k] = src.attn[pmc * (unsigned)LP + t];
            }
        }
#pragma unroll
        for (int i = 0; i < kStageIters; ++i) {
            const int idx = tid + i * kTileThreads;
            if (idx < n_stage)
                *reinterpret_cast<f32x4 *>(G + (unsigned)(idx >> 3) * kBinsGRow + (unsigned)(idx & 7) * 16u) =
                    st_ok[i] ? st_g[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            if (SOFT) {          // <grad_out_row, out_row>: eight consecutive lanes hold a row's eight chunks
                const f32x4 pr = st_g[i] * st_o[i];
                float d8 = (pr.x + pr.y) + (pr.z + pr.w);
                d8 = row_sum<8>(d8);
                if (idx < n_stage && (idx & 7) == 0)
                    reinterpret_cast<float *>(s_dyn + bp.o_dot)[idx >> 3] = st_ok[i] ? d8 : 0.f;
            }
        }
        if (SOFT) {     // (P = 4, L = 4) the items' softmax weights (msda_common.h: sm_exp / sm_rcp, adjacent-pair tree)
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                // lane p of the quad holds logits 4 p .. 4 p + 3 of its row: Q_p = (e0 + e1) + (e2 + e3), the sum is
                // (Q_0 + Q_1) + (Q_2 + Q_3) by the quad butterflies; this item's point is l P + p = logit p of lane l
                const f32x4 g4 = lg4[k];
                float mx = fmaxf(fmaxf(g4.x, g4.y), fmaxf(g4.z, g4.w));
                mx = fmaxf(mx, MSDA_DPP(mx, 0xB1));
                mx = fmaxf(mx, MSDA_DPP(mx, 0x4E));
                const float e0 = sm_exp(g4.x, mx), e1 = sm_exp(g4.y, mx), e2 = sm_exp(g4.z, mx), e3 = sm_exp(g4.w, mx);
                float sum = (e0 + e1) + (e2 + e3);
                sum += MSDA_DPP(sum, 0xB1);
                sum += MSDA_DPP(sum, 0x4E);
                const float rs = sm_rcp(sum);
                // quad lane l's four values, then component p of them
                const unsigned ql = (unsigned)l;
                auto bcast = [&](float v) {
                    const float b0 = MSDA_DPP(v, 0x00), b1 = MSDA_DPP(v, 0x55), b2 = MSDA_DPP(v, 0xAA), b3 = MSDA_DPP(v, 0xFF);
                    return ql == 0u ? b0 : (ql == 1u ? b1 : (ql == 2u ? b2 : b3));
                };
                const float c0 = bcast(e0), c1 = bcast(e1), c2 = bcast(e2), c3 = bcast(e3);
                const int it = tid + k * kTileThreads;
                const int itc = it < n_items ? it : 0;
                const int p = itc & 3;
                araw[k] = (p == 0 ? c0 : (p == 1 ? c1 : (p == 2 ? c2 : c3))) * rs;
            }
        }
        float sx = 0.f, sy = 0.f, cn = 0.f;
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const bool ok = okk[k];
            f32x2 xy = bp.fused_loc ? fused_location_from(raw[k], r01[k], r23[k], src.ref_dim, P, H, W) : raw[k];
            xy = ok ? xy : f32x2{0.f, 0.f};
            const float a = ok ? araw[k] : 0.f;
            const Sample<float> s = sample_setup<float>(xy.x, xy.y, H, W);
            const bool gate = s.gate && ok;
            it_live[k] = ok;                       // the row exists
            it_gate[k] = gate;
            it_h0[k] = s.h_low;
            it_w0[k] = s.w_low;
            it_lh[k] = gate ? s.lh : 0.f;
            it_lw[k] = gate ? s.lw : 0.f;
            it_a[k] = gate ? a : 0.f;
            it_araw[k] = a;                        // (soft: the softmax Jacobian wants the weight of gated-off points too)
            if (gate) {
                sx += (float)s.w_low + s.lw;
                sy += (float)s.h_low + s.lh;
                cn += 1.f;
            }
        }
        sx = bins_wave_sum(sx);
        sy = bins_wave_sum(sy);
        cn = bins_wave_sum(cn);
        if (lane == 0) {
            atomicAdd(&s_sum[0], sx);
            atomicAdd(&s_sum[1], sy);
            atomicAdd(&s_sum[2], cn);
        }
    }
    __syncthreads();      // B1: grad_out rows staged, position sums complete
    if (pl.ablate & 8) return;      // (profiling: bits 8 / 16 stop after phase 0 / 1, 32 / 64 skip the row phase / the sort)

    // ---- phase 1: window origin (every thread: the same inputs give the same bits), tickets, records ----
    int oy, ox, oyu, oxu;        // window origin, clamped to the level / as measured
    {
        const float cnt = s_sum[2];
        const float cx = cnt > 0.f ? s_sum[0] / cnt : (float)((rx << shl) + (1 << shl) / 2);
        const float cy = cnt > 0.f ? s_sum[1] / cnt : (float)((ry << shl) + (1 << shl) / 2);
        ox = (int)floorf(cx - 0.5f * (float)(win - 1) + 0.5f);
        oy = (int)floorf(cy - 0.5f * (float)(win - 1) + 0.5f);
        const int max_x = W - win, max_y = H - win;
        oxu = ox;
        oyu = oy;
        ox = ox > max_x ? max_x : ox;
        oy = oy > max_y ? max_y : oy;
        ox = ox < 0 ? 0 : ox;
        oy = oy < 0 ? 0 : oy;
    }
    const long mask_base = (long)b * pl.S + lstart_l;
    unsigned cr[NI][4];          // cell | ticket << 16 of the corners that are inside the window, else ~0
    float wa[NI][4];
    unsigned my_cnt = 0u;        // statistics: valid | outside the window << 10 | outside the shrunk window << 20
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int h0 = it_h0[k], w0 = it_w0[k];
        const bool on = it_gate[k];
        const bool okh0 = on && h0 >= 0, okh1 = on && h0 + 1 <= H - 1;
        const bool okw0 = w0 >= 0, okw1 = w0 + 1 <= W - 1;
        bool v[4] = {okh0 && okw0, okh0 && okw1, okh1 && okw0, okh1 && okw1};
        if (src.mask != nullptr) {       // (the split fused backward runs this kernel with the padding mask of `value`)
            const unsigned char *mk = src.mask + mask_base;
            const int p00 = bins_mul24(h0, W) + w0;
            v[0] = v[0] && !mk[v[0] ? p00 : 0];
            v[1] = v[1] && !mk[v[1] ? p00 + 1 : 0];
            v[2] = v[2] && !mk[v[2] ? p00 + W : 0];
            v[3] = v[3] && !mk[v[3] ? p00 + W + 1 : 0];
        }
        const int wy0 = h0 - oy, wx0 = w0 - ox;
        const bool iy0 = (unsigned)wy0 < (unsigned)win, iy1 = (unsigned)(wy0 + 1) < (unsigned)win;
        const bool ix0 = (unsigned)wx0 < (unsigned)win, ix1 = (unsigned)(wx0 + 1) < (unsigned)win;
        const bool in[4] = {v[0] && iy0 && ix0, v[1] && iy0 && ix1, v[2] && iy1 && ix0, v[3] && iy1 && ix1};
        const unsigned c00 = (unsigned)(bins_mul24(wy0, win) + wx0);
        const unsigned cell[4] = {c00, c00 + 1u, c00 + (unsigned)win, c00 + (unsigned)win + 1u};
        const float lh = it_lh[k], lw = it_lw[k], a = it_a[k];
        const float hh = 1.f - lh, hw = 1.f - lw;
        const float wk[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
        unsigned fl = (it_live[k] ? 0x100u : 0u) | (it_gate[k] ? 0x200u : 0u);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            cr[k][c] = 0xffffffffu;
            wa[k][c] = wk[c] * a;
            if (in[c]) {
                const unsigned ticket = atomicAdd(&CNT[cell[c]], 1u);
                cr[k][c] = cell[c] | (ticket << 16);
            }
            fl |= (v[c] ? 1u : 0u) << c;
            fl |= ((v[c] && !in[c]) ? 1u : 0u) << (4 + c);
        }
        if (stat_wg) {
            // the window a margin smaller by `shrink` would have had: same centre, clamped to the level the same way
            const int sh = bp.shrink, ws = win - 2 * sh;
            const int mxs = W - ws > 0 ? W - ws : 0, mys = H - ws > 0 ? H - ws : 0;
            int oxs = oxu + sh < mxs ? oxu + sh : mxs, oys = oyu + sh < mys ? oyu + sh : mys;
            oxs = oxs < 0 ? 0 : oxs;
            oys = oys < 0 ? 0 : oys;
            const unsigned wn = (unsigned)ws;
            const bool jy0 = (unsigned)(h0 - oys) < wn, jy1 = (unsigned)(h0 + 1 - oys) < wn;
            const bool jx0 = (unsigned)(w0 - oxs) < wn, jx1 = (unsigned)(w0 + 1 - oxs) < wn;
            const unsigned ni = (v[0] && !(jy0 && jx0) ? 1u : 0u) + (v[1] && !(jy0 && jx1) ? 1u : 0u) +
                                (v[2] && !(jy1 && jx0) ? 1u : 0u) + (v[3] && !(jy1 && jx1) ? 1u : 0u);
            my_cnt += (unsigned)__builtin_popcount(fl & 15u) + ((unsigned)__builtin_popcount((fl >> 4) & 15u) << 10) + (ni << 20);
        }
        const int it = tid + k * kTileThreads;
        if (it < n_items) {
            u32x4 rec;
            rec.x = (((unsigned)b * (unsigned)pl.S + (unsigned)(lstart_l + bins_mul24(h0, W) + w0)) * (unsigned)M + (unsigned)m) * ROWB;
            rec.y = __float_as_uint(lh);
            rec.z = __float_as_uint(lw);
            rec.w = __float_as_uint(it_araw[k]);
            R[it] = rec;
            FL[it] = fl;
        }
    }
    if (stat_wg) {       // (fields stay below 2^10: at most 64 lanes x NI x 4 corners)
        unsigned c = my_cnt;
        c += BINS_DPP_U(c, 0xB1);
        c += BINS_DPP_U(c, 0x4E);
        c += BINS_DPP_U(c, 0x141);
        c += BINS_DPP_U(c, 0x140);
        const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)c, 0), r1 = (unsigned)__builtin_amdgcn_readlane((int)c, 16);
        const unsigned r2 = (unsigned)__builtin_amdgcn_readlane((int)c, 32), r3 = (unsigned)__builtin_amdgcn_readlane((int)c, 48);
        if (lane == 0) {
            atomicAdd(&s_cnt[0], (r0 & 1023u) + (r1 & 1023u) + (r2 & 1023u) + (r3 & 1023u));
            atomicAdd(&s_cnt[1], ((r0 >> 10) & 1023u) + ((r1 >> 10) & 1023u) + ((r2 >> 10) & 1023u) + ((r3 >> 10) & 1023u));
            atomicAdd(&s_cnt[2], (r0 >> 20) + (r1 >> 20) + (r2 >> 20) + (r3 >> 20));
        }
    }
    __syncthreads();      // B2: tickets drawn, records written
    if (bp.stats != nullptr) {
        if (tid == 0 && stat_wg) sel_add(bp.stats, bp.level, (unsigned)(sw >> 3), s_cnt[0], s_cnt[1], s_cnt[2]);
        if (sw == 0 && wave == 1) sel_publish(bp.stats, bp.stats_host, lane);
    }
    if (pl.ablate & 16) return;

    // ---- phase 2a (the last wavefront, which has one row step fewer): prefix sum over the cells, list of the
    //      non-empty ones.  Touches only counters / START / COMP -- the records are still being read ----
    if (wave == kTileThreads / 64 - 1 && !(pl.ablate & 64)) {
        const int C = bp.scan_c;
        unsigned packed = 0u;          // entries in the low half, non-empty cells in the high half
        for (int j = 0; j < C; j += 4) {
            const u32x4 c4 = *reinterpret_cast<const u32x4 *>(&CNT[lane * C + j]);
            packed += (c4.x + c4.y + c4.z + c4.w) +
                      (((c4.x ? 1u : 0u) + (c4.y ? 1u : 0u) + (c4.z ? 1u : 0u) + (c4.w ? 1u : 0u)) << 16);
        }
        const unsigned incl = bins_incl_scan(packed, lane);
        unsigned run = incl - packed;
        for (int j = 0; j < C; j += 4) {
            const u32x4 c4 = *reinterpret_cast<const u32x4 *>(&CNT[lane * C + j]);
            unsigned s4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s4[e] = run & 0xffffu;
                if (c4[e] != 0u) {
                    COMP[run >> 16] = (unsigned)(lane * C + j + e) | (c4[e] << 16);
                    run += c4[e] + 0x10000u;
                }
            }
            *reinterpret_cast<u32x2 *>(&START[lane * C + j]) = u32x2{s4[0] | (s4[1] << 16), s4[2] | (s4[3] << 16)};
        }
        if (lane == 63) *reinterpret_cast<unsigned *>(s_dyn + bp.o_misc + 12u) = incl;
    }

    // ---- phase 2: the items' gradient dot products: 4 lanes x 8 channels per item, 16 items per wavefront step ----
    const int grp = lane >> 2, j4 = lane & 3;
    const unsigned ch_a = kB16 ? 2u * (unsigned)j4 : (unsigned)j4;             // this lane's two 16-byte chunks of a
    const unsigned ch_b = kB16 ? 2u * (unsigned)j4 + 1u : (unsigned)j4 + 4u;   // staged (fp32) grad_out row
    const unsigned ps = (unsigned)M * ROWB, wps = (unsigned)W * ps;
    const unsigned gps = (unsigned)M * 128u, gwps = (unsigned)W * gps;
    {
        // lane roles inside an item's quad: 0 -> d/dx, 1 -> d/dy, 2 -> d/d(attention), 3 -> none
        const bool role_y = j4 == 1, role_a = j4 == 2;
        const float size_r = bp.offsets_done ? 1.f : (role_y ? (float)H : (float)W);
        const bool split = grad_proj != nullptr;
        float *const dst_base = split ? grad_proj : (role_a ? grad_attn : grad_loc);
        const unsigned sh_row = (!split && !role_a) ? 1u : 0u;      // plain grad_loc rows are twice as long
        const unsigned sh_p = role_a ? 0u : 1u;
        const unsigned col = split ? (role_a ? (unsigned)(src.n_off + m * LP + l * P) : (unsigned)(m * 2 * LP + l * P * 2 + j4))
                                   : (role_a ? (unsigned)(l * P) : (unsigned)(l * P * 2 + j4));
        for (int st = wave; st * 16 < n_items && !(pl.ablate & 32); st += kTileThreads / 64) {
            const int it = st * 16 + grp;
            const bool vi = it < n_items;
            const int itc = vi ? it : n_items - 1;
            const u32x4 rec = R[itc];
            const unsigned fl = vi ? FL[itc] : 0u;
            const int r = bins_mul24(itc, bp.magic_p) >> 16, p = itc - bins_mul24(r, P);
            const unsigned char *const grow = G + (unsigned)r * kBinsGRow + ch_a * 16u;
            const f32x4 ga = *reinterpret_cast<const f32x4 *>(grow);
            const f32x4 gb = *reinterpret_cast<const f32x4 *>(grow + kChunkDelta);
            const unsigned rowa = ROWA[r];
            const unsigned base = rec.x + (unsigned)j4 * 16u;
            const bool ld = !(pl.ablate & 4);
            const unsigned off[4] = {((fl & 1u) && ld) ? base : kOobOffset, ((fl & 2u) && ld) ? base + ps : kOobOffset,
                                     ((fl & 4u) && ld) ? base + wps : kOobOffset,
                                     ((fl & 8u) && ld) ? base + wps + ps : kOobOffset};
            f32x4 va[4], vb[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (kB16) {
                    const u32x4 u = buf_load_u4(vr, off[k]);
                    va[k] = f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                                  __uint_as_float(u.y & 0xffff0000u)};
                    vb[k] = f32x4{__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16),
                                  __uint_as_float(u.w & 0xffff0000u)};
                } else {
                    va[k] = buf_load_f4(vr, off[k]);
                    vb[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(vr, (int)off[k], 64, 0));
                }
            }
            float d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                d[k] = bins_dot8(ga, va[k], gb, vb[k]);
                MSDA_QUAD_SUM(d[k]);
            }
            const float lh = __uint_as_float(rec.y), lw = __uint_as_float(rec.z), a_raw = __uint_as_float(rec.w);
            const float a = (fl & 0x200u) ? a_raw : 0.f;
            const float hh = 1.f - lh, hw = 1.f - lw;
            const float wk[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
            // d/dx = a (hh (d1 - d0) + lh (d3 - d2)) W,  d/dy = a (hw (d2 - d0) + lw (d3 - d1)) H: one expression, the
            // lane's role picks the operands (no divergent branches)
            const float pp = role_y ? hw : hh, qq = role_y ? lw : lh;
            const float d_a = role_y ? d[2] : d[1], d_c = role_y ? d[1] : d[2];
            const float r_loc = (a * size_r) * (pp * (d_a - d[0]) + qq * (d[3] - d_c));
            float r_att = wk[0] * d[0] + wk[1] * d[1] + wk[2] * d[2] + wk[3] * d[3];
            // soft: the final logit gradient a_t (ga_t - <grad_out_row, out_row>)
            if (SOFT) r_att = a_raw * (r_att - reinterpret_cast<const float *>(s_dyn + bp.o_dot)[r]);
            const float outv = role_a ? r_att : r_loc;
            if ((fl & 0x100u) && j4 < 3) dst_base[(rowa << sh_row) + col + ((unsigned)p << sh_p)] = outv;
            if (!(pl.ablate & 2) && __builtin_amdgcn_ballot_w64((fl & 0xf0u) != 0u) != 0ull) {   // rare: corners outside the window
                const unsigned gbase = rec.x * (128u / ROWB);
                const unsigned dg[4] = {0u, gps, gwps, gwps + gps};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (fl & (16u << k)) {
                        const float w = wk[k] * a;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(w * ga[i], gr, (int)(gbase + dg[k] + ch_a * 16u + (unsigned)i * 4u), 0, 0);
                            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(w * gb[i], gr, (int)(gbase + dg[k] + ch_b * 16u + (unsigned)i * 4u), 0, 0);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();      // B3: the records are dead -- the sorted entries take their place

    // ---- phase 3: the entries go to their places (START came from the last wavefront's scan, before B3) ----
    u32x2 *const E = reinterpret_cast<u32x2 *>(s_dyn + bp.o_x);
    unsigned *const s_tot = reinterpret_cast<unsigned *>(s_dyn + bp.o_misc + 12u);     // entries | non-empty cells << 16
    const unsigned last = (pl.ablate & 64) ? 0u : (unsigned)__builtin_amdgcn_readfirstlane((int)*s_tot);
    const unsigned total = last & 0xffffu, nz_total = last >> 16;
    if (!(pl.ablate & 64)) {
        if (tid == 0) E[total] = u32x2{(unsigned)rows * kBinsGRow, 0u};      // the list terminator: zero row, weight 0
#pragma unroll
        for (int k = 0; k < NI; ++k) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (cr[k][c] != 0xffffffffu) {
                    const unsigned pos = (unsigned)START[cr[k][c] & 0xffffu] + (cr[k][c] >> 16);
                    E[pos] = u32x2{(unsigned)it_r[k] * kBinsGRow, __float_as_uint(wa[k][c])};
                }
            }
        }
    }
    __syncthreads();      // B4: every entry is in place; counters and row tables are dead (the flush transpose takes their LDS)
    if (pl.ablate & 2) return;

    // ---- phase 4: per non-empty cell, gather its entries; flush the sums as whole 128-byte rows ----
    unsigned char *const ST = s_dyn + bp.o_st + (unsigned)wave * kBinsStageWave;
    const int half = lane >> 5, c32 = lane & 31;
    const unsigned sent = bp.o_x + total * 8u;                      // LDS offset of the terminator
    const unsigned g_a = ch_a * 16u, g_b = ch_b * 16u;
    // (tried in round 5, measured, dropped -- profiles/r05_lib_ab_bwd_*.txt: several quads per cell for workgroups with
    //  few non-empty cells, +3 us; the compacted list sorted by list length so that a wavefront's 16 cells have like trip
    //  counts -- 35 % fewer inner iterations (tools/gather_sim.py), -2 us fused / 0 plain, not worth its 10-bit counters;
    //  the four ticket atomics of an item issued without branches and the empty second item slot skipped per wavefront: 0)
    for (int round = 0; round * 64 + wave * 16 < (int)nz_total; ++round) {
        const int idx = round * 64 + wave * 16 + grp;
        const bool live = idx < (int)nz_total;
        const unsigned comp = live ? COMP[idx] : 0u;
        const unsigned cell = comp & 0xffffu, n = comp >> 16;
        unsigned pe = bp.o_x + (unsigned)START[cell] * 8u;
        f32x4 acc_a = f32x4{0.f, 0.f, 0.f, 0.f}, acc_b = f32x4{0.f, 0.f, 0.f, 0.f};
        // trip count: the longest list among this wavefront's 16 cells (the 4 lanes of a cell hold the same n)
        unsigned nm = n;
        nm = max(nm, BINS_DPP_U(nm, 0x124));      // row_ror:4
        nm = max(nm, BINS_DPP_U(nm, 0x128));      // row_ror:8
        const unsigned nmax = max(max((unsigned)__builtin_amdgcn_readlane((int)nm, 0), (unsigned)__builtin_amdgcn_readlane((int)nm, 16)),
                                  max((unsigned)__builtin_amdgcn_readlane((int)nm, 32), (unsigned)__builtin_amdgcn_readlane((int)nm, 48)));
        const unsigned lim = pe + n * 8u;          // one past this cell's last entry
        // (the entries of iteration i + 1 are requested before the rows of iteration i are used: one LDS round trip per
        //  iteration instead of two)
        u32x2 e0 = *reinterpret_cast<const u32x2 *>(s_dyn + (pe < lim ? pe : sent));
        u32x2 e1 = *reinterpret_cast<const u32x2 *>(s_dyn + (pe + 8u < lim ? pe + 8u : sent));
        for (unsigned i = 0; i < nmax; i += 2) {
            pe += 16u;
            const unsigned char *const g0 = G + e0.x + g_a, *const g1 = G + e1.x + g_a;
            const f32x4 x0 = *reinterpret_cast<const f32x4 *>(g0);
            const f32x4 y0 = *reinterpret_cast<const f32x4 *>(g0 + kChunkDelta);
            const f32x4 x1 = *reinterpret_cast<const f32x4 *>(g1);
            const f32x4 y1 = *reinterpret_cast<const f32x4 *>(g1 + kChunkDelta);
            const float w0 = __uint_as_float(e0.y), w1 = __uint_as_float(e1.y);
            e0 = *reinterpret_cast<const u32x2 *>(s_dyn + (pe < lim ? pe : sent));
            e1 = *reinterpret_cast<const u32x2 *>(s_dyn + (pe + 8u < lim ? pe + 8u : sent));
            __builtin_amdgcn_sched_barrier(0);      // (keep the requests above the FMAs: the scheduler sinks them otherwise)
            acc_a += w0 * x0;
            acc_b += w0 * y0;
            acc_a += w1 * x1;
            acc_b += w1 * y1;
        }
        if (pl.ablate & 1) continue;
        if (j4 == 0) {      // where the cell's row goes (slots past the list: dropped by the buffer's range check)
            const int wy = bins_mul24((int)cell, magic) >> 16, wx = (int)cell - bins_mul24(wy, win);
            const unsigned goff = (((unsigned)b * (unsigned)pl.S + (unsigned)(lstart_l + bins_mul24(oy + wy, W) + ox + wx)) *
                                   (unsigned)M + (unsigned)m) * 128u;
            s_off[grp] = live ? goff : kOobOffset;
        }
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if ((grp >> 3) == pass) {
                *reinterpret_cast<f32x4 *>(ST + (unsigned)(grp & 7) * 128u + g_a) = acc_a;
                *reinterpret_cast<f32x4 *>(ST + (unsigned)(grp & 7) * 128u + g_b) = acc_b;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            float fv[4];
            unsigned fo[4];
#pragma unroll
            for (int h = 0; h < 4; ++h) {       // (eight reads in flight, then the four rows leave)
                const int slot = 2 * h + half;
                fv[h] = *reinterpret_cast<const float *>(ST + (unsigned)slot * 128u + (unsigned)c32 * 4u);
                fo[h] = s_off[pass * 8 + slot] + (unsigned)c32 * 4u;      // (kOobOffset + 124: still out of range)
            }
#pragma unroll
            for (int h = 0; h < 4; ++h) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(fv[h], gr, (int)fo[h], 0, 0);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}
