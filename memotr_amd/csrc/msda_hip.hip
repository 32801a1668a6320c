// msda_hip.hip -- multi-scale deformable attention for gfx950 (MI355X, CDNA4).
//
// Hand-written HIP; wave64, LDS-staged sampling records, buffer (SRSRC) gathers with
// hardware zero padding, DPP reductions, fixed-point LDS accumulation, hardware f32 atomics.
// No CUDA-compat layer.
//
// Semantics replaced (reference repository paths):
//   forward   models/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299 (+ bilinear :33-84)
//   backward  models/ops/src/cuda/ms_deform_im2col_cuda.cuh:301-403 (+ bilinear :87-159)
//   host side models/ops/src/cuda/ms_deform_attn_cuda.cu:20-153
//   fused prologue (msda_fused_*): models/ops/modules/ms_deform_attn.py:104-123 -- softmax over the L*P
//             logits, sampling locations from offsets + reference points, padding-mask fill of `value`
// C ABI: include/msda_hip.h.  Design notes, byte counts and rooflines: DESIGN.md.
//
// Variant numbers (msda_set_option "fwd_variant" / "bwd_variant"; 0 = auto):
//   forward : 0 auto (fp32 pyramid self-attention with host shapes: 12 unless "fwd_win_auto" is 0; other D = 32
//             calls: 3) | 1 generic | 2,3,4 d32 gather with 2,4,1 points in flight | 8,9 region-tiled hybrid
//             (level 0 through the vector L1, coarser levels from LDS windows; 4 / 2 global points in flight; round 2,
//             slower) | 12 msda_fwd_d32_win (msda_fwd_win.h, round 3): (batch, head, 8x8-pixel region) per workgroup,
//             windows of levels 1-3 filled by LDS-DMA around the measured mean offset, level 0 through the vector L1,
//             pixel-pair LDS reads, one head per XCD
//   backward: 0 auto (pyramid self-attention: 10; other D = 32 calls: msda_bwd_d32_rows, 32 lanes per row) | 1 generic |
//             8,9 region-tiled fixed-point windows, all levels of a region per workgroup (2 / 4 points
//             in flight) | 10,11 region-tiled fixed-point windows, one pyramid level per workgroup, inputs loaded once
//
// Kernel families
//   *_generic   any D/L/P, f32 / f64 / bf16 storage: one thread per output scalar
//               (forward) or one block per (n,q,m) row (backward).  Correctness path for
//               shapes the specialised kernels do not cover (reference gradcheck sizes
//               D in {30,64,71,1025,...}).
//   *_d32_win / *_d32_tile_* / *_d32_rows   region- or row-organised specialisations, described at their definitions
//               (msda_fwd_win.h, below, msda_bwd_rows.h)
//   *_d32       MeMOTR geometry (D = 32 channels/head): the lanes that own one (n,q,m) row hold its
//               32 channels (8 lanes x 4 fp32 channels, 4 lanes x 8 bf16 channels).  Each lane prepares
//               the sampling record of a share of the row's L*P points exactly once, parks it in LDS, and the
//               lanes of the row then stream the records back as broadcast ds_read_b128.  Corner reads
//               are 16-byte buffer loads; invalid corners carry an out-of-range offset so the buffer unit
//               returns zeros (= the reference's per-corner zero padding, no divergent branches).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>

#include "../../include/msda_hip.h"
#include "msda_common.h"
#include "msda_select.h"

namespace {

using namespace msda;

template <typename T>
__device__ __forceinline__ void atomic_add_hw(T *p, T v) {
    unsafeAtomicAdd(p, v);  // global_atomic_add_f32 / _f64, no CAS loop
}

__device__ __forceinline__ float t_exp(float x) { return expf(x); }
__device__ __forceinline__ double t_exp(double x) { return exp(x); }

// ----------------------------------------------------------------------------------------
// generic forward: one thread per output scalar (n,q,m,c); consecutive threads walk c.
// FUSED (float only): locations / weights come from the raw projection + reference points (PointSrc).
// ----------------------------------------------------------------------------------------
template <typename TV, typename TC, bool FUSED>
__global__ __launch_bounds__(256) void msda_fwd_generic(
    const TV *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lstart,
    const TC *__restrict__ loc, const TC *__restrict__ attn, const PointSrc fs, int N, int S, int M, int D, int L,
    int Lq, int P, TV *__restrict__ out) {
    const long total = (long)N * Lq * M * D;
    const long row = (long)M * D;
    const int LP = L * P;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % D);
        const long pm = idx / D;
        const int m = (int)(pm % M);
        const long b = pm / M / Lq;
        const TC *lp = loc + pm * LP * 2;
        const TC *ap = attn + pm * LP;
        TC mx = 0, rsum = 1;
        const float *lg = nullptr;
        if constexpr (FUSED) {
            lg = fused_logits(fs, pm / M, m, LP);
            mx = lg[0];
            for (int t = 1; t < LP; ++t) mx = fmaxf(mx, lg[t]);
            TC sum = 0;
            for (int t = 0; t < LP; ++t) sum += t_exp(lg[t] - mx);
            rsum = (TC)1 / sum;
        }
        TC acc = (TC)0;
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            const TV *v = value + (b * S + lstart[l]) * row + (long)m * D + c;
            const unsigned char *msk = nullptr;
            if constexpr (FUSED) msk = fs.mask ? fs.mask + b * S + lstart[l] : nullptr;
            for (int p = 0; p < P; ++p) {
                TC lx, ly, a;
                if constexpr (FUSED) {
                    const int t = l * P + p;
                    const f32x2 xy = fused_location(fs, pm / M, m, L, P, t, l, H, W);
                    lx = xy.x;
                    ly = xy.y;
                    a = t_exp(lg[t] - mx) * rsum;
                } else {
                    lx = lp[0];
                    ly = lp[1];
                    a = ap[0];
                    lp += 2;
                    ap += 1;
                }
                const Sample<TC> s = sample_setup<TC>(lx, ly, H, W);
                if (!s.gate) continue;
                const TC hh = (TC)1 - s.lh, hw = (TC)1 - s.lw;
                const int h0 = s.h_low, w0 = s.w_low, h1 = h0 + 1, w1 = w0 + 1;
                TC v1 = 0, v2 = 0, v3 = 0, v4 = 0;
                if (h0 >= 0 && w0 >= 0 && !(msk && msk[h0 * W + w0])) v1 = to_compute(v[((long)h0 * W + w0) * row]);
                if (h0 >= 0 && w1 <= W - 1 && !(msk && msk[h0 * W + w1])) v2 = to_compute(v[((long)h0 * W + w1) * row]);
                if (h1 <= H - 1 && w0 >= 0 && !(msk && msk[h1 * W + w0])) v3 = to_compute(v[((long)h1 * W + w0) * row]);
                if (h1 <= H - 1 && w1 <= W - 1 && !(msk && msk[h1 * W + w1])) v4 = to_compute(v[((long)h1 * W + w1) * row]);
                const TC val = (hh * hw) * v1 + (hh * s.lw) * v2 + (s.lh * hw) * v3 + (s.lh * s.lw) * v4;
                acc += val * a;
            }
        }
        out[idx] = to_storage<TV, TC>(acc);
    }
}

// ----------------------------------------------------------------------------------------
// generic backward: one block per (n,q,m) row, threads stride over channels; per (l,p) the
// channel partials of grad_loc / grad_attn are reduced wave-wide with shuffles and across
// waves through LDS.  grad_value goes out as hardware atomics.
// FUSED: the per-point results stay in LDS and leave through the Jacobians of the prologue:
//   grad_logit_t = a_t (grad_attn_t - sum_j a_j grad_attn_j)          (softmax)
//   grad_off     = grad_loc / (W, H)      or   grad_loc * ref_wh * 0.5 / P
//   grad_ref     (optional, per head; the caller sums over heads): sum_p grad_loc, sum_p grad_loc * off * 0.5 / P
// ----------------------------------------------------------------------------------------
template <typename TV, typename TC, typename TG, bool FUSED>
__global__ __launch_bounds__(1024) void msda_bwd_generic(
    const TV *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lstart,
    const TC *__restrict__ loc, const TC *__restrict__ attn, const PointSrc fs, const TV *__restrict__ grad_out, int N,
    int S, int M, int D, int L, int Lq, int P, TG *__restrict__ grad_value, TC *__restrict__ grad_loc,
    TC *__restrict__ grad_attn, float *__restrict__ grad_proj, float *__restrict__ grad_ref_part) {
    __shared__ TC red[3 * 16];
    __shared__ float s_res[FUSED ? 3 * kMaxFusedLP : 1];
    const long n_rows = (long)N * Lq * M;
    const long row = (long)M * D;
    const int LP = L * P;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    for (long pm = blockIdx.x; pm < n_rows; pm += gridDim.x) {
        const int m = (int)(pm % M);
        const long b = pm / M / Lq;
        const TV *g = grad_out + pm * D;
        TC mx = 0, rsum = 1;
        const float *lg = nullptr;
        if constexpr (FUSED) {
            lg = fused_logits(fs, pm / M, m, LP);
            mx = lg[0];
            for (int t = 1; t < LP; ++t) mx = fmaxf(mx, lg[t]);
            TC sum = 0;
            for (int t = 0; t < LP; ++t) sum += t_exp(lg[t] - mx);
            rsum = (TC)1 / sum;
        }
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            const long base = (b * S + lstart[l]) * row + (long)m * D;
            const unsigned char *msk = nullptr;
            if constexpr (FUSED) msk = fs.mask ? fs.mask + b * S + lstart[l] : nullptr;
            for (int p = 0; p < P; ++p) {
                const long t = (pm * L + l) * P + p;
                TC lx, ly, a;
                if constexpr (FUSED) {
                    const int tt = l * P + p;
                    const f32x2 xy = fused_location(fs, pm / M, m, L, P, tt, l, H, W);
                    lx = xy.x;
                    ly = xy.y;
                    a = t_exp(lg[tt] - mx) * rsum;
                } else {
                    lx = loc[2 * t];
                    ly = loc[2 * t + 1];
                    a = attn[t];
                }
                const Sample<TC> s = sample_setup<TC>(lx, ly, H, W);
                TC acc_w = 0, acc_h = 0, acc_a = 0;
                if (s.gate) {  // block-uniform
                    const TC hh = (TC)1 - s.lh, hw = (TC)1 - s.lw;
                    const TC w1 = hh * hw, w2 = hh * s.lw, w3 = s.lh * hw, w4 = s.lh * s.lw;
                    const int h0 = s.h_low, w0 = s.w_low, h1 = h0 + 1, w1i = w0 + 1;
                    bool ok1 = (h0 >= 0 && w0 >= 0), ok2 = (h0 >= 0 && w1i <= W - 1);
                    bool ok3 = (h1 <= H - 1 && w0 >= 0), ok4 = (h1 <= H - 1 && w1i <= W - 1);
                    if (msk) {   // padded pixels: value reads as 0 and receives no gradient (masked_fill)
                        ok1 = ok1 && !msk[h0 * W + w0];
                        ok2 = ok2 && !msk[h0 * W + w1i];
                        ok3 = ok3 && !msk[h1 * W + w0];
                        ok4 = ok4 && !msk[h1 * W + w1i];
                    }
                    const long i1 = base + ((long)h0 * W + w0) * row, i2 = base + ((long)h0 * W + w1i) * row;
                    const long i3 = base + ((long)h1 * W + w0) * row, i4 = base + ((long)h1 * W + w1i) * row;
                    for (int c = threadIdx.x; c < D; c += blockDim.x) {
                        const TC top = to_compute(g[c]);
                        const TC tga = top * a;
                        TC v1 = 0, v2 = 0, v3 = 0, v4 = 0;
                        if (ok1) { v1 = to_compute(value[i1 + c]); atomic_add_hw<TG>(grad_value + i1 + c, (TG)(w1 * tga)); }
                        if (ok2) { v2 = to_compute(value[i2 + c]); atomic_add_hw<TG>(grad_value + i2 + c, (TG)(w2 * tga)); }
                        if (ok3) { v3 = to_compute(value[i3 + c]); atomic_add_hw<TG>(grad_value + i3 + c, (TG)(w3 * tga)); }
                        if (ok4) { v4 = to_compute(value[i4 + c]); atomic_add_hw<TG>(grad_value + i4 + c, (TG)(w4 * tga)); }
                        const TC gw = hh * (v2 - v1) + s.lh * (v4 - v3);
                        const TC gh = hw * (v3 - v1) + s.lw * (v4 - v2);
                        acc_a += top * (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
                        acc_w += (TC)W * gw * tga;
                        acc_h += (TC)H * gh * tga;
                    }
                }
                acc_w = wave_sum(acc_w);
                acc_h = wave_sum(acc_h);
                acc_a = wave_sum(acc_a);
                if (n_waves > 1) {
                    if (lane == 0) {
                        red[wave * 3 + 0] = acc_w;
                        red[wave * 3 + 1] = acc_h;
                        red[wave * 3 + 2] = acc_a;
                    }
                    __syncthreads();
                    if (threadIdx.x == 0) {
                        for (int w = 1; w < n_waves; ++w) {
                            acc_w += red[w * 3 + 0];
                            acc_h += red[w * 3 + 1];
                            acc_a += red[w * 3 + 2];
                        }
                    }
                }
                if (threadIdx.x == 0) {
                    if constexpr (FUSED) {
                        const int tt = l * P + p;
                        s_res[2 * tt] = (float)acc_w;
                        s_res[2 * tt + 1] = (float)acc_h;
                        s_res[2 * LP + tt] = (float)acc_a;
                    } else {
                        grad_loc[2 * t] = acc_w;
                        grad_loc[2 * t + 1] = acc_h;
                        grad_attn[t] = acc_a;
                    }
                }
                if (n_waves > 1) __syncthreads();
            }
        }
        if constexpr (FUSED) {
            __syncthreads();
            const long qrow = pm / M;
            const int tid = threadIdx.x;
            float *gp = grad_proj + qrow * fs.proj_stride;
            if (tid < LP) {
                float dot = 0.f;
                for (int j = 0; j < LP; ++j) dot += (expf(lg[j] - mx) * rsum) * s_res[2 * LP + j];
                const float a_t = expf(lg[tid] - mx) * rsum;
                gp[fs.n_off + m * LP + tid] = a_t * (s_res[2 * LP + tid] - dot);
                const int l = tid / P;
                const float *r = fs.ref + (qrow * L + l) * fs.ref_dim;
                float jx, jy;
                if (fs.ref_dim == 2) {
                    jx = 1.f / (float)shapes[2 * l + 1];
                    jy = 1.f / (float)shapes[2 * l];
                    gp[(m * LP + tid) * 2] = s_res[2 * tid] / (float)shapes[2 * l + 1];
                    gp[(m * LP + tid) * 2 + 1] = s_res[2 * tid + 1] / (float)shapes[2 * l];
                } else {
                    jx = r[2] * (0.5f / (float)P);
                    jy = r[3] * (0.5f / (float)P);
                    gp[(m * LP + tid) * 2] = s_res[2 * tid] * jx;
                    gp[(m * LP + tid) * 2 + 1] = s_res[2 * tid + 1] * jy;
                }
                (void)jx; (void)jy;
            }
            if (grad_ref_part != nullptr && tid < L * fs.ref_dim) {
                const int l = tid / fs.ref_dim, comp = tid - l * fs.ref_dim;
                const float *off = fs.proj + qrow * fs.proj_stride + ((long)m * LP + l * P) * 2;
                float acc = 0.f;
                for (int p = 0; p < P; ++p) {
                    const float gl = s_res[2 * (l * P + p) + (comp & 1)];
                    acc += comp < 2 ? gl : gl * off[2 * p + (comp & 1)] * (0.5f / (float)P);
                }
                grad_ref_part[(pm * L + l) * fs.ref_dim + comp] = acc;
            }
            __syncthreads();
        }
    }
}

// ----------------------------------------------------------------------------------------
// parity hooks: the shared sample_setup, and the fused prologue (locations + softmax weights) as the
// kernels compute them.
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void msda_indices_f32_kernel(const int64_t *__restrict__ shapes,
                                                              const float *__restrict__ loc, long n_points, int L,
                                                              int P, int32_t *__restrict__ h_low,
                                                              int32_t *__restrict__ w_low, uint8_t *__restrict__ gate) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n_points; t += (long)gridDim.x * blockDim.x) {
        const int l = (int)((t / P) % L);
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const Sample<float> s = sample_setup<float>(loc[2 * t], loc[2 * t + 1], H, W);
        h_low[t] = s.h_low;
        w_low[t] = s.w_low;
        gate[t] = s.gate ? 1 : 0;
    }
}

// one wavefront handles 8 rows (8 lanes each), exactly like the specialised kernels' staging step
__global__ __launch_bounds__(256) void msda_fused_points_kernel(const int64_t *__restrict__ shapes, const PointSrc fs,
                                                               long n_rows, int M, int L, int P,
                                                               float *__restrict__ loc_out,
                                                               float *__restrict__ attn_out) {
    const int LP = L * P;
    const int sub = threadIdx.x & 7;
    for (long pm0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 3; pm0 < ((n_rows + 7) & ~7L);
         pm0 += ((long)gridDim.x * blockDim.x) >> 3) {
        const bool ok = pm0 < n_rows;
        const long pm = ok ? pm0 : n_rows - 1;
        const int m = (int)(pm % M);
        const float *lg = fused_logits(fs, pm / M, m, LP);
        float mx, rsum;
        row_softmax_stats<8>(lg, LP, sub, mx, rsum);
        for (int t = sub; t < LP; t += 8) {
            const int l = t / P;
            const f32x2 xy = fused_location(fs, pm / M, m, L, P, t, l, (int)shapes[2 * l], (int)shapes[2 * l + 1]);
            if (ok) {
                loc_out[(pm * LP + t) * 2] = xy.x;
                loc_out[(pm * LP + t) * 2 + 1] = xy.y;
                attn_out[pm * LP + t] = expf(lg[t] - mx) * rsum;
            }
        }
    }
}

// The same for L*P <= 16 with one lane per point (16 lanes per row): every lane reads and writes consecutive
// addresses.  The softmax sum adds in the order of the 8-lane form above ((t, t+8) pairs first, then the butterfly
// over 8 lanes), so both produce the same bits.
__global__ __launch_bounds__(256) void msda_fused_points16_kernel(const int64_t *__restrict__ shapes, const PointSrc fs,
                                                                 long n_rows, int M, int L, int P,
                                                                 float *__restrict__ loc_out,
                                                                 float *__restrict__ attn_out) {
    const int LP = L * P;
    const int t = threadIdx.x & 15;
    const long rows_pad = (n_rows + 3) & ~3L;
    for (long pm0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4; pm0 < rows_pad;
         pm0 += ((long)gridDim.x * blockDim.x) >> 4) {
        const bool ok = pm0 < n_rows && t < LP;
        const long pm = pm0 < n_rows ? pm0 : n_rows - 1;
        const long qrow = pm / M;
        const int m = (int)(pm - qrow * M);
        const float lg = t < LP ? fused_logits(fs, qrow, m, LP)[t] : -INFINITY;
        float mx = lg;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
        const float e = expf(lg - mx);
        float sum = e + __shfl_xor(e, 8, 16);
        sum += __shfl_xor(sum, 1, 16);
        sum += __shfl_xor(sum, 2, 16);
        sum += __shfl_xor(sum, 4, 16);
        const float rsum = 1.f / sum;
        if (ok) {
            if (loc_out != nullptr) {     // (null: the consumer computes the locations itself, msda_bwd_bins.h)
                const int l = t / P;
                const f32x2 xy = fused_location(fs, qrow, m, L, P, t, l, (int)shapes[2 * l], (int)shapes[2 * l + 1]);
                *reinterpret_cast<f32x2 *>(loc_out + (pm * LP + t) * 2) = xy;
            }
            attn_out[pm * LP + t] = e * rsum;
        }
    }
}

// ----------------------------------------------------------------------------------------
// D = 32 specialised kernels.
// ----------------------------------------------------------------------------------------
// Row geometry by storage type: a row (one pixel of one head) is 32 channels = 128 B (fp32) or 64 B (bf16); every
// lane moves 16 bytes per corner, so 8 (fp32) or 4 (bf16) lanes own a row and a wavefront owns 8 or 16 rows.
template <typename TV>
struct RowGeom {
    static constexpr int kRowBytes = 32 * (int)sizeof(TV);
    static constexpr int kLanes = kRowBytes / 16;      // lanes per row
    static constexpr int kCh = 32 / kLanes;            // channels per lane
    static constexpr int kRows = 64 / kLanes;          // rows per wavefront
};

// acc[0..kCh) += w * (16 bytes of a row)
template <typename TV>
__device__ __forceinline__ void fma_row16(float *acc, float w, const u32x4 v);
template <>
__device__ __forceinline__ void fma_row16<float>(float *acc, float w, const u32x4 v) {
    acc[0] += w * __uint_as_float(v.x);
    acc[1] += w * __uint_as_float(v.y);
    acc[2] += w * __uint_as_float(v.z);
    acc[3] += w * __uint_as_float(v.w);
}
template <>
__device__ __forceinline__ void fma_row16<bf16_t>(float *acc, float w, const u32x4 v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        acc[2 * i] += w * __uint_as_float(v[i] << 16);
        acc[2 * i + 1] += w * __uint_as_float(v[i] & 0xffff0000u);
    }
}

template <typename TV>
__device__ __forceinline__ void store_row16(TV *dst, const float *acc);
template <>
__device__ __forceinline__ void store_row16<float>(float *dst, const float *acc) {
    *reinterpret_cast<f32x4 *>(dst) = f32x4{acc[0], acc[1], acc[2], acc[3]};
}
template <>
__device__ __forceinline__ void store_row16<bf16_t>(bf16_t *dst, const float *acc) {
    u32x4 p;
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = bf16_bits_rne(acc[2 * i]) | (bf16_bits_rne(acc[2 * i + 1]) << 16);
    *reinterpret_cast<u32x4 *>(dst) = p;
}

// Prepare the sampling records this lane owns for its row and park them in LDS: 32 bytes per (row, point) =
// 4 corner byte offsets (kOobOffset when the corner is outside the level, masked, or the point is gated off) +
// the 4 bilinear corner weights pre-multiplied by the attention weight.
template <typename TV, bool FUSED>
__device__ __forceinline__ void stage_records_fwd(u32x4 *rec, const PointSrc &src, unsigned pmc, unsigned qrow, int m,
                                                  bool row_ok, int sub, int L, int P, int M, int S, int b,
                                                  unsigned row_base, const int *s_H, const int *s_W,
                                                  const int *s_start) {
    constexpr int LANES = RowGeom<TV>::kLanes;
    constexpr unsigned ROWB = RowGeom<TV>::kRowBytes;
    const int LP = L * P;
    float mx = 0.f, rsum = 1.f;
    const float *lg = nullptr;
    float e0 = 0.f, e1 = 0.f;            // exp(logit - max) of this lane's first two points (all of them when LP <= 2 LANES)
    const bool two = LP <= 2 * LANES;
    if (FUSED) {
        lg = fused_logits(src, qrow, m, LP);
        if (two) {
            const float l0 = sub < LP ? lg[sub] : -INFINITY, l1 = sub + LANES < LP ? lg[sub + LANES] : -INFINITY;
            mx = row_max<LANES>(fmaxf(l0, l1));
            e0 = expf(l0 - mx);
            e1 = expf(l1 - mx);
            rsum = 1.f / row_sum<LANES>(e0 + e1);
        } else {
            row_softmax_stats<LANES>(lg, LP, sub, mx, rsum);
        }
    }
    const float rcp_p = 1.f / (float)P;
    for (int t = sub; t < LP; t += LANES) {
        const int l = (int)(((float)t + 0.5f) * rcp_p);      // == t / P (the product stays 0.5/P away from integers)
        const int H = s_H[l], W = s_W[l];
        const f32x2 xy = point_location<FUSED>(src, pmc, qrow, m, L, P, t, l, H, W);
        const float a_in = FUSED ? (two ? (t == sub ? e0 : e1) : expf(lg[t] - mx)) * rsum
                                 : src.attn[pmc * (unsigned)LP + (unsigned)t];
        Sample<float> s = sample_setup<float>(xy.x, xy.y, H, W);
        const bool live = s.gate && row_ok;
        // a gated-off point contributes nothing (the reference skips it): no NaN * 0 from non-finite locations
        const float a = live ? a_in : 0.f;
        if (!s.gate) s.lh = s.lw = 0.f;
        const float hh = 1.f - s.lh, hw = 1.f - s.lw;
        const int h0 = s.h_low, w0 = s.w_low, h1 = h0 + 1, w1 = w0 + 1;
        const bool okh0 = live && h0 >= 0, okh1 = live && h1 <= H - 1;
        const bool okw0 = w0 >= 0, okw1 = w1 <= W - 1;
        bool ok00 = okh0 && okw0, ok01 = okh0 && okw1, ok10 = okh1 && okw0, ok11 = okh1 && okw1;
        if (FUSED && src.mask != nullptr) {
            const unsigned char *mk = src.mask + ((unsigned)b * (unsigned)S + (unsigned)s_start[l]);
            const int p00 = h0 * W + w0;
            ok00 = ok00 && !mk[ok00 ? p00 : 0];
            ok01 = ok01 && !mk[ok01 ? p00 + 1 : 0];
            ok10 = ok10 && !mk[ok10 ? p00 + W : 0];
            ok11 = ok11 && !mk[ok11 ? p00 + W + 1 : 0];
        }
        const unsigned pix_stride = (unsigned)M * ROWB;
        const unsigned o00 = row_base + (unsigned)(s_start[l] + h0 * W + w0) * pix_stride;
        u32x4 off;
        off.x = ok00 ? o00 : kOobOffset;
        off.y = ok01 ? o00 + pix_stride : kOobOffset;
        off.z = ok10 ? o00 + (unsigned)W * pix_stride : kOobOffset;
        off.w = ok11 ? o00 + (unsigned)W * pix_stride + pix_stride : kOobOffset;
        f32x4 w;
        w.x = (hh * hw) * a;
        w.y = (hh * s.lw) * a;
        w.z = (s.lh * hw) * a;
        w.w = (s.lh * s.lw) * a;
        rec[2 * t] = off;
        rec[2 * t + 1] = __builtin_bit_cast(u32x4, w);
    }
}

// One chunk of PTS points of one row: all 4*PTS corner loads are issued before the first
// FMA so a wave keeps 4*PTS 16-byte-per-lane requests in flight.
template <int PTS, typename TV>
__device__ __forceinline__ void fwd_gather_chunk(const u32x4 *rec, int t0, __amdgpu_buffer_rsrc_t vr,
                                                 unsigned lane_off, float *acc) {
    u32x4 o[PTS];
    f32x4 w[PTS];
    u32x4 v[PTS][4];
#pragma unroll
    for (int i = 0; i < PTS; ++i) {
        o[i] = rec[2 * (t0 + i)];
        w[i] = __builtin_bit_cast(f32x4, rec[2 * (t0 + i) + 1]);
    }
#pragma unroll
    for (int i = 0; i < PTS; ++i) {
        v[i][0] = buf_load_u4(vr, o[i].x + lane_off);
        v[i][1] = buf_load_u4(vr, o[i].y + lane_off);
        v[i][2] = buf_load_u4(vr, o[i].z + lane_off);
        v[i][3] = buf_load_u4(vr, o[i].w + lane_off);
    }
#pragma unroll
    for (int i = 0; i < PTS; ++i) {
        fma_row16<TV>(acc, w[i].x, v[i][0]);
        fma_row16<TV>(acc, w[i].y, v[i][1]);
        fma_row16<TV>(acc, w[i].z, v[i][2]);
        fma_row16<TV>(acc, w[i].w, v[i][3]);
    }
}

// forward, variants 2/3/4: direct gather (every corner row is read through the vector L1).
// PTS = points whose corner loads are kept in flight together.
template <int PTS, typename TV, bool FUSED>
__global__ __launch_bounds__(256) void msda_fwd_d32_gather(
    const TV *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lstart,
    const PointSrc src, int N, int S, int M, int L, int Lq, int P, TV *__restrict__ out, unsigned value_bytes,
    int head_major) {
    constexpr int D = 32;
    constexpr int LANES = RowGeom<TV>::kLanes, ROWS = RowGeom<TV>::kRows, CH = RowGeom<TV>::kCh;
    __shared__ int s_H[kMaxLevels], s_W[kMaxLevels], s_start[kMaxLevels];
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    if (threadIdx.x < L) {
        s_H[threadIdx.x] = (int)shapes[2 * threadIdx.x];
        s_W[threadIdx.x] = (int)shapes[2 * threadIdx.x + 1];
        s_start[threadIdx.x] = (int)lstart[threadIdx.x];
    }
    __syncthreads();
    const int LP = L * P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int grp = lane / LANES, sub = lane % LANES;
    const int rec_stride = 2 * LP + 1;  // in 16-byte units; +1 staggers the rows over LDS banks
    u32x4 *rec = reinterpret_cast<u32x4 *>(s_dyn) + (size_t)(wave * ROWS + grp) * rec_stride;
    // 32-bit row arithmetic: the launch envelope (check_dims) keeps every element index below 2^31
    const unsigned n_rows = (unsigned)N * (unsigned)Lq * (unsigned)M;
    // head-major walk (option fwd_head_major): a wavefront owns ROWS consecutive queries of ONE head and the XCDs split
    // the heads, so each XCD's L2 holds one head's slab of `value` instead of a band of all heads
    const unsigned n_q = (unsigned)N * (unsigned)Lq, q_tasks = (n_q + ROWS - 1) / ROWS;
    const unsigned n_tasks = head_major ? q_tasks * (unsigned)M : (n_rows + ROWS - 1) / ROWS;
    const __amdgpu_buffer_rsrc_t vr = make_rsrc(value, value_bytes);
    const unsigned lane_off = (unsigned)sub * 16u;
    const TaskWalk tw = xcd_walk(n_tasks, wpb);
    for (long task = tw.begin; task < tw.end; task += tw.step) {
        unsigned pm;
        bool row_ok;
        if (head_major) {
            const unsigned hm = (unsigned)task / q_tasks, qq = ((unsigned)task - hm * q_tasks) * ROWS + grp;
            row_ok = qq < n_q;
            pm = (row_ok ? qq : n_q - 1) * (unsigned)M + hm;
        } else {
            pm = (unsigned)task * ROWS + grp;
            row_ok = pm < n_rows;
        }
        const unsigned pmc = row_ok ? pm : n_rows - 1;
        const unsigned qrow = pmc / (unsigned)M;
        const int m = (int)(pmc - qrow * (unsigned)M);
        const int b = (int)(qrow / (unsigned)Lq);
        const unsigned row_base = ((unsigned)b * (unsigned)S * (unsigned)M + (unsigned)m) * (D * (unsigned)sizeof(TV));
        stage_records_fwd<TV, FUSED>(rec, src, pmc, qrow, m, row_ok, sub, L, P, M, S, b, row_base, s_H, s_W, s_start);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float acc[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) acc[i] = 0.f;
        int t = 0;
        for (; t + PTS <= LP; t += PTS) fwd_gather_chunk<PTS, TV>(rec, t, vr, lane_off, acc);
        for (; t < LP; ++t) fwd_gather_chunk<1, TV>(rec, t, vr, lane_off, acc);
        if (row_ok) store_row16<TV>(out + (pm * (unsigned)D + (unsigned)(sub * CH)), acc);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ----------------------------------------------------------------------------------------
// Region-tiled kernels for self-attention over the pyramid (one query per pixel, Lq == S).
//
// A workgroup owns one (batch, region, head).  A region is the set of queries whose pixels fall
// in one cell of the coarsest level's grid: side_l = 2^(L-1-l) pixels per side at level l
// (8x8 + 4x4 + 2x2 + 1 = 85 queries for L = 4).  All of them sample the same neighbourhood of
// every level, so the workgroup keeps one window of this head's rows per level in LDS
// (128 B per pixel), centred on the mean sampling position it measures first:
//   forward (hybrid): windows of the coarser levels hold `value`; their corner reads are ds_read_b128
//             (LDS: 256 B/clk/CU) while the finest level keeps going through the vector L1 (64 B/clk/CU),
//             so the two pipes work side by side
//   backward: windows hold the grad_value partial sums as fixed point; corner scatters are LDS integer
//             atomics and the windows are flushed once with coalesced global float atomics
// Corners that fall outside a window take the global path (buffer load / buffer atomic), so
// results do not depend on where the samples are -- only the speed does.
// ----------------------------------------------------------------------------------------
constexpr int kTileMaxL = 4;
constexpr int kTileThreads = 256;
constexpr int kTileMaxRows = 85;          // 64 + 16 + 4 + 1

struct TilePlan {
    int N, S, M, L, P, Lq;
    int RY, RX;
    int rows;                      // queries per region
    int l0;                        // first level that has an LDS window (forward hybrid); 0 = all levels
    int wide_log2;                 // tiled backward: rows of a region differing by >= 2^wide_log2 make it "wide" (0 = never)
    int ablate;                    // profiling only (msda_set_option "bwd_ablate"): 1 no flush, 2 no scatter, 4 no value loads
    int H[kTileMaxL], W[kTileMaxL];
    int qstart[kTileMaxL];         // first query of level l (cumulative H*W)
    int shift[kTileMaxL];          // log2(side_l)
    int row0[kTileMaxL + 1];       // first region-row of level l
    int win[kTileMaxL];            // window side in pixels (0: no window)
    int win_magic[kTileMaxL];      // (x * magic) >> 16 == x / win for x < win*win
    int win_base[kTileMaxL + 1];   // first window pixel of level l (cumulative, pixels)
    unsigned value_bytes;
    int n_blocks;                  // real block count (grid is padded to a multiple of 8)
};

struct TileTables {  // LDS copy of the per-level tables (divergent lookups)
    int H[kTileMaxL], W[kTileMaxL], qstart[kTileMaxL], shift[kTileMaxL], row0[kTileMaxL + 1];
    int win[kTileMaxL], magic[kTileMaxL], base[kTileMaxL + 1], lstart[kTileMaxL];
    int oy[kTileMaxL], ox[kTileMaxL];
    float sum[kTileMaxL][3];
};

struct TileRow {
    bool ok;
    int q;       // query index inside the batch element
    long pm;     // (b*Lq + q)*M + m
};

__device__ __forceinline__ TileRow tile_row(const TileTables &tb, int L, int rows, int r, int b, int ry, int rx,
                                            int m, int M, int Lq) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < kTileMaxL; ++i)
        if (i < L && r >= tb.row0[i]) l = i;
    const int local = r - tb.row0[l], sh = tb.shift[l];
    const int py = (ry << sh) + (local >> sh), px = (rx << sh) + (local & ((1 << sh) - 1));
    TileRow o;
    o.ok = (r < rows) && (py < tb.H[l]) && (px < tb.W[l]);
    o.q = o.ok ? tb.qstart[l] + py * tb.W[l] + px : 0;
    o.pm = ((long)b * Lq + o.q) * M + m;
    return o;
}

__device__ __forceinline__ void tile_block_coords(const TilePlan &pl, int &b, int &ry, int &rx, int &m, bool &live) {
    // XCD-aware: block i runs on XCD i % 8; hand each XCD a contiguous run of (region, head) pairs
    const int nb_pad = gridDim.x, chunk = nb_pad >> 3;
    const int sw = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    live = sw < pl.n_blocks;
    const int id = live ? sw : 0;
    m = id % pl.M;
    const int reg = (id / pl.M) % (pl.RY * pl.RX);
    b = id / (pl.M * pl.RY * pl.RX);
    ry = reg / pl.RX;
    rx = reg - ry * pl.RX;
}

__device__ __forceinline__ void tile_load_tables(TileTables &tb, const TilePlan &pl,
                                                 const int64_t *__restrict__ lstart) {
    const int t = threadIdx.x;
    if (t < kTileMaxL) {
        tb.H[t] = pl.H[t];
        tb.W[t] = pl.W[t];
        tb.qstart[t] = pl.qstart[t];
        tb.shift[t] = pl.shift[t];
        tb.win[t] = pl.win[t];
        tb.magic[t] = pl.win_magic[t];
        tb.lstart[t] = t < pl.L ? (int)lstart[t] : 0;
        tb.sum[t][0] = tb.sum[t][1] = tb.sum[t][2] = 0.f;
        tb.oy[t] = tb.ox[t] = 0;
    }
    if (t <= kTileMaxL) {
        tb.row0[t] = pl.row0[t];
        tb.base[t] = pl.win_base[t];
    }
}

// Measure the mean sampling position of every windowed level over the region's gated points and place
// the windows around it.  Ends with a __syncthreads(); tb.oy/ox are valid afterwards.
template <bool FUSED>
__device__ __forceinline__ void tile_place_windows(TileTables &tb, const TilePlan &pl, const PointSrc &src, int b,
                                                   int ry, int rx, int m) {
    const int lane = threadIdx.x & 63;
    for (int l = pl.l0; l < pl.L; ++l) {
        float sx = 0.f, sy = 0.f, cnt = 0.f;
        const int n = pl.rows * pl.P;
        const int H = tb.H[l], W = tb.W[l];
        for (int i = threadIdx.x; i < n; i += kTileThreads) {
            const int r = i / pl.P, p = i - r * pl.P;
            const TileRow row = tile_row(tb, pl.L, pl.rows, r, b, ry, rx, m, pl.M, pl.Lq);
            if (row.ok) {
                const f32x2 xy = point_location<FUSED>(src, row.pm, (long)b * pl.Lq + row.q, m, pl.L, pl.P,
                                                       l * pl.P + p, l, H, W);
                const Sample<float> s = sample_setup<float>(xy.x, xy.y, H, W);
                if (s.gate) {
                    sx += (float)s.w_low + s.lw;
                    sy += (float)s.h_low + s.lh;
                    cnt += 1.f;
                }
            }
        }
        sx = wave_sum(sx);
        sy = wave_sum(sy);
        cnt = wave_sum(cnt);
        if (lane == 0) {
            atomicAdd(&tb.sum[l][0], sx);
            atomicAdd(&tb.sum[l][1], sy);
            atomicAdd(&tb.sum[l][2], cnt);
        }
    }
    __syncthreads();
    if (threadIdx.x >= pl.l0 && threadIdx.x < pl.L) {
        const int l = threadIdx.x, win = tb.win[l], sh = tb.shift[l];
        const float cnt = tb.sum[l][2];
        // no gated point at this level: centre on the region itself
        const float cx = cnt > 0.f ? tb.sum[l][0] / cnt : (float)((rx << sh) + (1 << sh) / 2);
        const float cy = cnt > 0.f ? tb.sum[l][1] / cnt : (float)((ry << sh) + (1 << sh) / 2);
        int ox = (int)floorf(cx - 0.5f * (float)(win - 1) + 0.5f);
        int oy = (int)floorf(cy - 0.5f * (float)(win - 1) + 0.5f);
        const int max_x = tb.W[l] - win, max_y = tb.H[l] - win;
        ox = ox > max_x ? max_x : ox;
        oy = oy > max_y ? max_y : oy;
        tb.ox[l] = ox < 0 ? 0 : ox;
        tb.oy[l] = oy < 0 ? 0 : oy;
    }
    __syncthreads();
}

// Byte offset of pixel (gy, gx) of level l, head m, batch b, relative to the tensor base (rows of ROWB bytes).
template <unsigned ROWB = 128u>
__device__ __forceinline__ unsigned tile_pixel_off(const TileTables &tb, const TilePlan &pl, int b, int l, int gy,
                                                   int gx, int m) {
    return (((unsigned)b * (unsigned)pl.S + (unsigned)(tb.lstart[l] + gy * tb.W[l] + gx)) * (unsigned)pl.M +
            (unsigned)m) * ROWB;
}

// validity of the four corners of a sample (inside the level, point gated on, pixel not padded)
struct Corners {
    bool v00, v01, v10, v11;
};
template <bool FUSED>
__device__ __forceinline__ Corners tile_corners(const Sample<float> &s, bool live, int H, int W, const PointSrc &src,
                                                long mask_base) {
    const int h0 = s.h_low, w0 = s.w_low, h1 = h0 + 1, w1 = w0 + 1;
    const bool on = s.gate && live;
    const bool okh0 = on && h0 >= 0, okh1 = on && h1 <= H - 1;
    const bool okw0 = w0 >= 0, okw1 = w1 <= W - 1;
    Corners c;
    c.v00 = okh0 && okw0;
    c.v01 = okh0 && okw1;
    c.v10 = okh1 && okw0;
    c.v11 = okh1 && okw1;
    if (FUSED && src.mask != nullptr) {
        const unsigned char *mk = src.mask + mask_base;
        const int p00 = h0 * W + w0;
        c.v00 = c.v00 && !mk[c.v00 ? p00 : 0];
        c.v01 = c.v01 && !mk[c.v01 ? p00 + 1 : 0];
        c.v10 = c.v10 && !mk[c.v10 ? p00 + W : 0];
        c.v11 = c.v11 && !mk[c.v11 ? p00 + W + 1 : 0];
    }
    return c;
}

// ---- forward, hybrid -----------------------------------------------------------------------------
// Record of one (row, point): 16 bytes.
//   word 0   byte offset of corner (h0, w0) -- inside the LDS windows when kRecLds is set, else inside `value`.
//            Both are multiples of 128, so the low 7 bits carry flags: bits 0-3 = corner validity, bit 4 = kRecLds.
//   words 1-3  lh, lw, attention weight
// A point of a windowed level whose valid corners are not ALL inside the window is recorded as a global point.
constexpr unsigned kRecLds = 16u;

template <int PTS>
__device__ __forceinline__ void hybrid_global_chunk(const u32x4 *rec, int t0, unsigned ps, unsigned wps,
                                                    __amdgpu_buffer_rsrc_t vr, unsigned lane_off, f32x4 &acc) {
    u32x4 r[PTS];
    f32x4 v[PTS][4];
#pragma unroll
    for (int i = 0; i < PTS; ++i) r[i] = rec[t0 + i];
#pragma unroll
    for (int i = 0; i < PTS; ++i) {
        const unsigned fl = r[i].x, base = (fl & ~127u) + lane_off;
        v[i][0] = buf_load_f4(vr, (fl & 1u) ? base : kOobOffset);
        v[i][1] = buf_load_f4(vr, (fl & 2u) ? base + ps : kOobOffset);
        v[i][2] = buf_load_f4(vr, (fl & 4u) ? base + wps : kOobOffset);
        v[i][3] = buf_load_f4(vr, (fl & 8u) ? base + wps + ps : kOobOffset);
    }
#pragma unroll
    for (int i = 0; i < PTS; ++i) {
        const float lh = __uint_as_float(r[i].y), lw = __uint_as_float(r[i].z), a = __uint_as_float(r[i].w);
        const float hw = 1.f - lw;
        const float ha = (1.f - lh) * a, la = lh * a;
        acc += (ha * hw) * v[i][0];
        acc += (ha * lw) * v[i][1];
        acc += (la * hw) * v[i][2];
        acc += (la * lw) * v[i][3];
    }
}

// PTS points of a windowed level: all 4*PTS ds_read_b128 are issued back to back (LDS latency is ~100 cycles: one
// point at a time leaves the pipe idle); a point that left its window drags the chunk through the slow path.
template <int PTS>
__device__ __forceinline__ void hybrid_lds_chunk(const u32x4 *rec, int t0, unsigned wrow, unsigned ps, unsigned wps,
                                                 unsigned zero_row, __amdgpu_buffer_rsrc_t vr, unsigned lane_off,
                                                 const unsigned char *s_dyn, f32x4 &acc) {
    u32x4 r[PTS];
    f32x4 v[PTS][4];
    bool any_global = false;
#pragma unroll
    for (int i = 0; i < PTS; ++i) {
        r[i] = rec[t0 + i];
        any_global = any_global || (!(r[i].x & kRecLds) && (r[i].x & 15u) != 0u);
    }
    const unsigned zr = zero_row + lane_off;
    if (__builtin_amdgcn_ballot_w64(any_global) == 0ull) {
#pragma unroll
        for (int i = 0; i < PTS; ++i) {
            const unsigned fl = r[i].x, base = (fl & ~127u) + lane_off;
            v[i][0] = *reinterpret_cast<const f32x4 *>(s_dyn + ((fl & 1u) ? base : zr));
            v[i][1] = *reinterpret_cast<const f32x4 *>(s_dyn + ((fl & 2u) ? base + 128u : zr));
            v[i][2] = *reinterpret_cast<const f32x4 *>(s_dyn + ((fl & 4u) ? base + wrow : zr));
            v[i][3] = *reinterpret_cast<const f32x4 *>(s_dyn + ((fl & 8u) ? base + wrow + 128u : zr));
        }
    } else {   // rare: some row's point left its window -> that lane takes the global path for it
        const unsigned dl[4] = {0u, 128u, wrow, wrow + 128u};
        const unsigned dg[4] = {0u, ps, wps, wps + ps};
#pragma unroll
        for (int i = 0; i < PTS; ++i) {
            const unsigned fl = r[i].x, base = (fl & ~127u) + lane_off;
            const bool in_lds = (fl & kRecLds) != 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool ok = (fl >> k) & 1u;
                v[i][k] = *reinterpret_cast<const f32x4 *>(s_dyn + ((ok && in_lds) ? base + dl[k] : zr));
                if (ok && !in_lds) v[i][k] = buf_load_f4(vr, base + dg[k]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < PTS; ++i) {
        const float lh = __uint_as_float(r[i].y), lw = __uint_as_float(r[i].z), a = __uint_as_float(r[i].w);
        const float hw = 1.f - lw;
        const float ha = (1.f - lh) * a, la = lh * a;
        acc += (ha * hw) * v[i][0];
        acc += (ha * lw) * v[i][1];
        acc += (la * hw) * v[i][2];
        acc += (la * lw) * v[i][3];
    }
}

// Round-2 structure (the one that fixed the backward, section 4.2 of DESIGN.md): every global input of the
// workgroup -- the locations / weights (or offsets, logits, reference points) of all its rows -- is loaded ONCE, up
// front, for the three 32-row passes and kept in registers (2 points per lane and pass: L*P <= 16); the window
// placement is computed from those registers, the windows are filled, and the passes then stage their records from
// registers: 5 dependent global round trips per workgroup instead of 13.
template <int PTS, bool FUSED>
__global__ __launch_bounds__(kTileThreads, 4) void msda_fwd_d32_hybrid(
    const float *__restrict__ value, const int64_t *__restrict__ lstart, const PointSrc src,
    float *__restrict__ out, const TilePlan pl) {
    constexpr int D = 32;
    constexpr int NP = (kTileMaxRows + 31) / 32;
    __shared__ TileTables tb;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    int b, ry, rx, m;
    bool live;
    tile_block_coords(pl, b, ry, rx, m, live);
    if (!live) return;
    tile_load_tables(tb, pl, lstart);
    __syncthreads();

    const int L = pl.L, P = pl.P, LP = L * P, l0 = pl.l0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 3, sub = lane & 7;
    const __amdgpu_buffer_rsrc_t vr = make_rsrc(value, pl.value_bytes);

    // ---- phase A: all locations / weights of the workgroup's rows, once ----
    bool ok[NP];
    long pmr[NP];
    float lx[NP][2], ly[NP][2], la[NP][2];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const TileRow row = tile_row(tb, L, pl.rows, p * 32 + wave * 8 + grp, b, ry, rx, m, pl.M, pl.Lq);
        ok[p] = row.ok;
        pmr[p] = row.pm;
        const long qrow = (long)b * pl.Lq + row.q;
        float mx = 0.f, rsum = 1.f, e[2] = {0.f, 0.f};
        if (FUSED) {     // softmax of the row's (<= 16) logits: two per lane, DPP reductions over the row's 8 lanes
            const float *lg = fused_logits(src, qrow, m, LP);
            const float g0 = sub < LP ? lg[sub] : -INFINITY, g1 = sub + 8 < LP ? lg[sub + 8] : -INFINITY;
            mx = row_max<8>(fmaxf(g0, g1));
            e[0] = expf(g0 - mx);
            e[1] = expf(g1 - mx);
            rsum = 1.f / row_sum<8>(e[0] + e[1]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int t = sub + 8 * j;
            lx[p][j] = ly[p][j] = la[p][j] = 0.f;
            if (row.ok && t < LP) {
                const int l = t / P;
                const f32x2 xy = point_location<FUSED>(src, row.pm, qrow, m, L, P, t, l, tb.H[l], tb.W[l]);
                lx[p][j] = xy.x;
                ly[p][j] = xy.y;
                la[p][j] = FUSED ? e[j] * rsum : src.attn[row.pm * LP + t];
            }
        }
    }

    // ---- phase B: mean sampling position of every windowed level, from registers ----
    {
        float sx[kTileMaxL] = {0.f, 0.f, 0.f, 0.f}, sy[kTileMaxL] = {0.f, 0.f, 0.f, 0.f}, sc[kTileMaxL] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int t = sub + 8 * j;
                if (ok[p] && t < LP) {
                    const int l = t / P;
                    const float Hf = (float)tb.H[l], Wf = (float)tb.W[l];
                    const float w_im = lx[p][j] * Wf - 0.5f, h_im = ly[p][j] * Hf - 0.5f;
                    const bool gate = h_im > -1.f && w_im > -1.f && h_im < Hf && w_im < Wf;
#pragma unroll
                    for (int k = 0; k < kTileMaxL; ++k)
                        if (gate && k == l) { sx[k] += w_im; sy[k] += h_im; sc[k] += 1.f; }
                }
            }
        for (int l = l0; l < L; ++l) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int k = 0; k < kTileMaxL; ++k)
                if (k == l) { a0 = sx[k]; a1 = sy[k]; a2 = sc[k]; }
            a0 = wave_sum(a0);
            a1 = wave_sum(a1);
            a2 = wave_sum(a2);
            if (lane == 0) {
                atomicAdd(&tb.sum[l][0], a0);
                atomicAdd(&tb.sum[l][1], a1);
                atomicAdd(&tb.sum[l][2], a2);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x >= l0 && threadIdx.x < L) {
        const int l = threadIdx.x, win = tb.win[l], sh = tb.shift[l];
        const float cnt = tb.sum[l][2];
        const float cx = cnt > 0.f ? tb.sum[l][0] / cnt : (float)((rx << sh) + (1 << sh) / 2);
        const float cy = cnt > 0.f ? tb.sum[l][1] / cnt : (float)((ry << sh) + (1 << sh) / 2);
        int ox = (int)floorf(cx - 0.5f * (float)(win - 1) + 0.5f);
        int oy = (int)floorf(cy - 0.5f * (float)(win - 1) + 0.5f);
        const int max_x = tb.W[l] - win, max_y = tb.H[l] - win;
        ox = ox > max_x ? max_x : ox;
        oy = oy > max_y ? max_y : oy;
        tb.ox[l] = ox < 0 ? 0 : ox;
        tb.oy[l] = oy < 0 ? 0 : oy;
    }
    __syncthreads();

    f32x4 *win_f4 = reinterpret_cast<f32x4 *>(s_dyn);
    const int win_px = tb.base[kTileMaxL];
    const unsigned zero_row = (unsigned)win_px * 128u;   // one all-zero pixel row after the windows
    const int rec_stride = LP + 1;                       // 16-byte units; +1 staggers the 8 rows over the banks
    u32x4 *rec = reinterpret_cast<u32x4 *>(s_dyn + (size_t)(win_px + 1) * 128) + (size_t)(wave * 8 + grp) * rec_stride;

    // ---- fill the windows (coalesced 128-byte rows; out-of-level / padded cells read as zero) ----
    if (threadIdx.x < 8) win_f4[win_px * 8 + threadIdx.x] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int l = l0; l < L; ++l) {
        const int win = tb.win[l], magic = tb.magic[l], base = tb.base[l];
        const int oy = tb.oy[l], ox = tb.ox[l], H = tb.H[l], W = tb.W[l];
        const int n = win * win * 8;
        const unsigned char *mk = (FUSED && src.mask != nullptr) ? src.mask + (long)b * pl.S + tb.lstart[l] : nullptr;
#pragma unroll 4
        for (int i = threadIdx.x; i < n; i += kTileThreads) {
            const int pix = i >> 3, s8 = i & 7;
            const int wy = (pix * magic) >> 16, wx = pix - wy * win;
            const int gy = oy + wy, gx = ox + wx;
            bool inside = gy < H && gx < W;
            if (mk != nullptr && inside) inside = !mk[gy * W + gx];
            const unsigned off = inside ? tile_pixel_off(tb, pl, b, l, gy, gx, m) + (unsigned)s8 * 16u : kOobOffset;
            win_f4[(base + pix) * 8 + s8] = buf_load_f4(vr, off);
        }
    }
    __syncthreads();

    // ---- phase C: the passes; records come from registers ----
    const unsigned lane_off = (unsigned)sub * 16u;
    const unsigned ps = (unsigned)pl.M * 128u;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (p * 32 >= pl.rows) break;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int t = sub + 8 * j;
            if (t < LP) {
                const int l = t / P;
                const int H = tb.H[l], W = tb.W[l];
                Sample<float> s = sample_setup<float>(lx[p][j], ly[p][j], H, W);
                const float a = (s.gate && ok[p]) ? la[p][j] : 0.f;   // gated-off points contribute nothing
                if (!s.gate) s.lh = s.lw = 0.f;
                const Corners c = tile_corners<FUSED>(s, ok[p], H, W, src, (long)b * pl.S + tb.lstart[l]);
                const unsigned valid = (unsigned)c.v00 | ((unsigned)c.v01 << 1) | ((unsigned)c.v10 << 2) | ((unsigned)c.v11 << 3);
                unsigned word0 = tile_pixel_off(tb, pl, b, l, s.h_low, s.w_low, m) | valid;
                if (l >= l0) {
                    const int win = tb.win[l];
                    const int wy0 = s.h_low - tb.oy[l], wx0 = s.w_low - tb.ox[l];
                    const bool iy0 = (unsigned)wy0 < (unsigned)win, iy1 = (unsigned)(wy0 + 1) < (unsigned)win;
                    const bool ix0 = (unsigned)wx0 < (unsigned)win, ix1 = (unsigned)(wx0 + 1) < (unsigned)win;
                    const bool all_in = (!c.v00 || (iy0 && ix0)) && (!c.v01 || (iy0 && ix1)) &&
                                        (!c.v10 || (iy1 && ix0)) && (!c.v11 || (iy1 && ix1));
                    if (all_in) word0 = ((unsigned)(tb.base[l] + wy0 * win + wx0) * 128u) | valid | kRecLds;
                }
                u32x4 r;
                r.x = word0;
                r.y = __float_as_uint(s.lh);
                r.z = __float_as_uint(s.lw);
                r.w = __float_as_uint(a);
                rec[t] = r;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // ---- levels read through the vector L1 ----
        for (int l = 0; l < l0; ++l) {
            const unsigned wps = (unsigned)__builtin_amdgcn_readfirstlane(tb.W[l]) * ps;
            int q = 0;
            for (; q + PTS <= P; q += PTS) hybrid_global_chunk<PTS>(rec, l * P + q, ps, wps, vr, lane_off, acc);
            for (; q < P; ++q) hybrid_global_chunk<1>(rec, l * P + q, ps, wps, vr, lane_off, acc);
        }
        // ---- levels read from the LDS windows ----
        for (int l = l0; l < L; ++l) {
            const unsigned wrow = (unsigned)__builtin_amdgcn_readfirstlane(tb.win[l]) * 128u;
            const unsigned wps = (unsigned)__builtin_amdgcn_readfirstlane(tb.W[l]) * ps;
            int q = 0;
            for (; q + PTS <= P; q += PTS)
                hybrid_lds_chunk<PTS>(rec, l * P + q, wrow, ps, wps, zero_row, vr, lane_off, s_dyn, acc);
            for (; q < P; ++q) hybrid_lds_chunk<1>(rec, l * P + q, wrow, ps, wps, zero_row, vr, lane_off, s_dyn, acc);
        }
        if (ok[p]) *reinterpret_cast<f32x4 *>(out + pmr[p] * D + sub * 4) = acc;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

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

template <int PTS, typename TV, bool FUSED>
__global__ __launch_bounds__(kTileThreads, PTS <= 2 ? 4 : 2) void msda_bwd_d32_tile_q2(
    const TV *__restrict__ value, const int64_t *__restrict__ lstart, const PointSrc src,
    const TV *__restrict__ grad_out, float *__restrict__ grad_value, float *__restrict__ grad_loc,
    float *__restrict__ grad_attn, float *__restrict__ grad_proj, const TilePlan pl) {
    constexpr int D = 32;
    constexpr unsigned ROWB = 32u * (unsigned)sizeof(TV);       // bytes of one value row
    __shared__ TileTables tb;
    __shared__ unsigned s_gbits[D];            // max |grad_out| bit pattern per channel over the region
    __shared__ unsigned s_abits[kTileMaxL];    // max |attention| bit pattern per level
    __shared__ float s_cscale[D], s_cinv[D];   // 2^(K - gexp[c]) and its inverse
    __shared__ float s_lscale[kTileMaxL], s_linv[kTileMaxL];
    __shared__ int s_nonfinite;
    __shared__ unsigned s_rowrange[2];         // min / max over the region's rows of max_c |grad_out[row, c]| (bits)
    __shared__ f32x2 s_stat[FUSED ? kTileMaxRows + 11 : 1];     // softmax (max, 1/sum) of every row of the region
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    int b, ry, rx, m;
    bool live;
    tile_block_coords(pl, b, ry, rx, m, live);
    if (!live) return;
    tile_load_tables(tb, pl, lstart);
    if (threadIdx.x < D) s_gbits[threadIdx.x] = 0u;
    if (threadIdx.x < kTileMaxL) s_abits[threadIdx.x] = 0u;
    if (threadIdx.x == 0) {
        s_nonfinite = 0;
        s_rowrange[0] = 0x7f800000u;
        s_rowrange[1] = 0u;
    }
    __syncthreads();

    const int L = pl.L, P = pl.P, LP = L * P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 3, sub = lane & 7;
    const __amdgpu_buffer_rsrc_t vr = make_rsrc(value, pl.value_bytes);
    const __amdgpu_buffer_rsrc_t gr = make_rsrc(grad_value, (unsigned)((size_t)pl.N * pl.S * pl.M * D * 4u));
    u32x4 *win_u4 = reinterpret_cast<u32x4 *>(s_dyn);
    const int win_px = tb.base[kTileMaxL];
    const int rec_stride = 2 * LP + 1;
    // layout: [windows | 8 dump rows (one per row slot) | records]
    const unsigned dump_cell = (unsigned)(win_px + grp);
    u32x4 *rec = reinterpret_cast<u32x4 *>(s_dyn + (size_t)(win_px + 8) * 128) + (size_t)(wave * 8 + grp) * rec_stride;

    for (int i = threadIdx.x; i < (win_px + 8) * 8; i += kTileThreads) win_u4[i] = u32x4{0u, 0u, 0u, 0u};
    {   // ---- bounds of this region: max |grad_out| per channel, max attention per level ----
        // 8 lanes per row, 32 rows per pass (the same row <-> lane map as the main loop)
        unsigned g4[4] = {0u, 0u, 0u, 0u};
        unsigned rmin = 0x7f800000u, rmax = 0u;
        unsigned a4[kTileMaxL] = {0u, 0u, 0u, 0u};
        for (int r0 = 0; r0 < pl.rows; r0 += 32) {
            const int r = r0 + (threadIdx.x >> 3);
            const TileRow row = tile_row(tb, L, pl.rows, r, b, ry, rx, m, pl.M, pl.Lq);
            unsigned rowm = 0u;
            if (row.ok) {
                const f32x4 g = load_ch4<TV>(grad_out + row.pm * D + sub * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned u = __float_as_uint(g[j]) & 0x7fffffffu;
                    g4[j] = u > g4[j] ? u : g4[j];
                    rowm = u > rowm ? u : rowm;
                }
            }
            // max over the 8 lanes of the row (non-negative floats order like their bit patterns)
            rowm = __float_as_uint(row_max<8>(__uint_as_float(rowm < 0x7f800000u ? rowm : 0x7f7fffffu)));
            if (row.ok && rowm != 0u) {
                rmin = rowm < rmin ? rowm : rmin;
                rmax = rowm > rmax ? rowm : rmax;
            }
            if (FUSED) {
                const float *lg = fused_logits(src, (long)b * pl.Lq + row.q, m, LP);
                float mx, rsum;
                row_softmax_stats<8>(lg, LP, sub, mx, rsum);
                if (sub == 0 && r < pl.rows) s_stat[r] = f32x2{mx, rsum};
                if (row.ok) {
                    for (int t = sub; t < LP; t += 8) {
                        const unsigned u = __float_as_uint(expf(lg[t] - mx) * rsum) & 0x7fffffffu;
                        const int l = t / P;
#pragma unroll
                        for (int k = 0; k < kTileMaxL; ++k) a4[k] = (k == l && u > a4[k]) ? u : a4[k];
                    }
                }
            } else if (row.ok) {
                for (int t = sub; t < LP; t += 8) {
                    const unsigned u = __float_as_uint(src.attn[row.pm * LP + t]) & 0x7fffffffu;
                    const int l = t / P;
#pragma unroll
                    for (int k = 0; k < kTileMaxL; ++k) a4[k] = (k == l && u > a4[k]) ? u : a4[k];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicMax(&s_gbits[sub * 4 + j], g4[j]);
#pragma unroll
        for (int k = 0; k < kTileMaxL; ++k)
            if (a4[k]) atomicMax(&s_abits[k], a4[k]);
        if (sub == 0 && rmax != 0u) {
            atomicMin(&s_rowrange[0], rmin);
            atomicMax(&s_rowrange[1], rmax);
        }
    }
    tile_place_windows<FUSED>(tb, pl, src, b, ry, rx, m);  // ends with __syncthreads()

    int cnt_log2 = 0;
    while ((1 << cnt_log2) < pl.rows * P) ++cnt_log2;
    int K = 30 - cnt_log2;
    K = K > 21 ? 21 : (K < 0 ? 0 : K);
    if (threadIdx.x < D + kTileMaxL) {
        const unsigned bits = threadIdx.x < D ? s_gbits[threadIdx.x] : s_abits[threadIdx.x - D];
        if (bits >= 0x7e800000u) atomicOr(&s_nonfinite, 1);      // inf / nan (or > 2^126): no fixed point here
    }
    __syncthreads();
    const bool nonfinite = s_nonfinite != 0;
    if (nonfinite) K = 0;
    if (threadIdx.x < D) {
        const int e = nonfinite ? 0 : bound_exponent(s_gbits[threadIdx.x]);
        s_cscale[threadIdx.x] = ldexpf(1.f, K - e);
        s_cinv[threadIdx.x] = ldexpf(1.f, e - K);
    } else if (threadIdx.x < D + kTileMaxL) {
        const int l = threadIdx.x - D;
        const int e = nonfinite ? 0 : bound_exponent(s_abits[l]);
        s_lscale[l] = ldexpf(1.f, -e);
        s_linv[l] = ldexpf(1.f, e);
    }
    __syncthreads();

    const unsigned lane_off = (unsigned)sub * (ROWB / 8u);          // this lane's 4 channels inside a value row
    const unsigned ps = (unsigned)pl.M * ROWB;
    // this lane scatters channel pairs {2 sub, 2 sub + 1} and {2 sub + 16, 2 sub + 17}: the 8 lanes of a row write
    // 64 contiguous bytes per ds_add_u64; the pair order alternates with the row slot to spread the LDS banks
    const int rot = grp & 1;
    const int cpair[2] = {2 * sub + 16 * rot, 2 * sub + 16 * (rot ^ 1)};
    // A region whose rows differ by >= 2^wide_log2 in magnitude ("wide": an outlier query, a dead neighbourhood) would
    // quantise its small rows at the large rows' step.  There, every lane whose own channels sit more than 7 bits
    // under their bounds sends its contributions through the float path instead; ordinary regions never take that test
    // (measured on the DanceTrack train step: with the threshold at 2^5 most regions of a real gradient qualify and the
    // kernel runs 9x slower; at the default 2^12 none do).
    const bool wide = pl.wide_log2 > 0 && s_rowrange[1] != 0u &&
                      (int)(s_rowrange[1] >> 23) - (int)(s_rowrange[0] >> 23) >= pl.wide_log2;
    const float lane_limit = wide ? ldexpf(1.f, K - 7) : 0.f;
    float cs[4], ci[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        cs[j] = s_cscale[cpair[j >> 1] + (j & 1)];
        ci[j] = s_cinv[cpair[j >> 1] + (j & 1)];
    }

    for (int r0 = 0; r0 < pl.rows; r0 += 32) {
        const int r = r0 + wave * 8 + grp;
        const TileRow row = tile_row(tb, L, pl.rows, r, b, ry, rx, m, pl.M, pl.Lq);
        const long qrow = (long)b * pl.Lq + row.q;
        const f32x4 g = row.ok ? load_ch4<TV>(grad_out + row.pm * D + sub * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        float gs[4];     // grad_out at the scatter channels of this lane, scaled to fixed-point units
#pragma unroll
        for (int j = 0; j < 4; ++j)
            gs[j] = row.ok ? to_compute(grad_out[row.pm * D + cpair[j >> 1] + (j & 1)]) * cs[j] : 0.f;
        // float path for this lane's channels?  (non-finite region: everything; wide region: small channels)
        bool lane_bypass = nonfinite;
#pragma unroll
        for (int j = 0; j < 4; ++j) lane_bypass = lane_bypass || (fabsf(gs[j]) < lane_limit && gs[j] != 0.f);
        const unsigned dump_or = lane_bypass ? 0xffffffffu : 0u;
        float mx = 0.f, rsum = 1.f;
        const float *lg = nullptr;
        if (FUSED) {
            lg = fused_logits(src, qrow, m, LP);
            const f32x2 st = s_stat[r < pl.rows ? r : 0];
            mx = st.x;
            rsum = st.y;
        }
        for (int t = sub; t < LP; t += 8) {
            const int l = t / P;
            const int H = tb.H[l], W = tb.W[l];
            const f32x2 xy = point_location<FUSED>(src, row.pm, qrow, m, L, P, t, l, H, W);
            const float a_in = FUSED ? expf(lg[t] - mx) * rsum : src.attn[row.pm * LP + t];
            Sample<float> s = sample_setup<float>(xy.x, xy.y, H, W);
            const float a = (s.gate && row.ok) ? a_in : 0.f;
            if (!s.gate) s.lh = s.lw = 0.f;
            const Corners c = tile_corners<FUSED>(s, row.ok, H, W, src, (long)b * pl.S + tb.lstart[l]);
            const int win = tb.win[l];
            const int wy0 = s.h_low - tb.oy[l], wx0 = s.w_low - tb.ox[l];
            const bool iy0 = (unsigned)wy0 < (unsigned)win, iy1 = (unsigned)(wy0 + 1) < (unsigned)win;
            const bool ix0 = (unsigned)wx0 < (unsigned)win, ix1 = (unsigned)(wx0 + 1) < (unsigned)win;
            const unsigned c00 = (unsigned)(tb.base[l] + wy0 * win + wx0);
            const bool w00 = c.v00 && iy0 && ix0, w01 = c.v01 && iy0 && ix1;
            const bool w10 = c.v10 && iy1 && ix0, w11 = c.v11 && iy1 && ix1;
            u32x4 r0v;
            r0v.x = tile_pixel_off<ROWB>(tb, pl, b, l, s.h_low, s.w_low, m);
            r0v.y = (w00 ? c00 : dump_cell) | ((w01 ? c00 + 1u : dump_cell) << 16);
            r0v.z = (w10 ? c00 + (unsigned)win : dump_cell) | ((w11 ? c00 + (unsigned)win + 1u : dump_cell) << 16);
            r0v.w = (unsigned)c.v00 | ((unsigned)c.v01 << 1) | ((unsigned)c.v10 << 2) | ((unsigned)c.v11 << 3) |
                    ((unsigned)(c.v00 && !w00) << 4) | ((unsigned)(c.v01 && !w01) << 5) |
                    ((unsigned)(c.v10 && !w10) << 6) | ((unsigned)(c.v11 && !w11) << 7) |
                    ((unsigned)(W * pl.M) << 8);
            f32x4 w;
            w.x = s.lh;
            w.y = s.lw;
            w.z = a;
            w.w = a * s_lscale[l];
            rec[2 * t] = r0v;
            rec[2 * t + 1] = __builtin_bit_cast(u32x4, w);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int t0 = 0; t0 < LP; t0 += PTS) {
            u32x4 ra[PTS];
            f32x4 rw[PTS], v[PTS][4];
#pragma unroll
            for (int i = 0; i < PTS; ++i) {
                const int t = (t0 + i < LP) ? t0 + i : LP - 1;
                ra[i] = rec[2 * t];
                rw[i] = __builtin_bit_cast(f32x4, rec[2 * t + 1]);
            }
#pragma unroll
            for (int i = 0; i < PTS; ++i) {
                const unsigned fl = ra[i].w, base = ra[i].x + lane_off;
                const unsigned wps = (fl >> 8) * ROWB;
                const bool ld = !(pl.ablate & 4);
                v[i][0] = buf_load_ch4<TV>(vr, ((fl & 1u) && ld) ? base : kOobOffset);
                v[i][1] = buf_load_ch4<TV>(vr, ((fl & 2u) && ld) ? base + ps : kOobOffset);
                v[i][2] = buf_load_ch4<TV>(vr, ((fl & 4u) && ld) ? base + wps : kOobOffset);
                v[i][3] = buf_load_ch4<TV>(vr, ((fl & 8u) && ld) ? base + wps + ps : kOobOffset);
            }
#pragma unroll
            for (int i = 0; i < PTS; ++i) {
                const int t = t0 + i;
                if (t < LP) {
                    const unsigned fl = ra[i].w;
                    const float lh = rw[i].x, lw = rw[i].y, a = rw[i].z, a_s = rw[i].w;
                    const float hh = 1.f - lh, hw = 1.f - lw;
                    const float wk[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
                    unsigned cells[4] = {ra[i].y & 0xffffu, ra[i].y >> 16, ra[i].z & 0xffffu, ra[i].z >> 16};
#pragma unroll
                    for (int k = 0; k < 4; ++k) cells[k] = dump_or ? dump_cell : cells[k];   // bypassing lane: dump row
                    const unsigned fpath = (fl >> 4) | (dump_or & fl);      // corners this lane sends as float atomics
                    // ---- grad_value: branch-free fixed-point scatter into the windows (or the dump row) ----
                    if (!(pl.ablate & 2))
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        unsigned char *p = s_dyn + (cells[k] << 7);
                        const float wa = wk[k] * a_s;
                        lds_add_pair(p + cpair[0] * 4, wa, gs[0], gs[1]);
                        lds_add_pair(p + cpair[1] * 4, wa, gs[2], gs[3]);
                    }
                    if (!(pl.ablate & 2) && __builtin_amdgcn_ballot_w64((fpath & 0xfu) != 0u) != 0ull) {   // rare: float path
                        const float li = s_linv[t / P];
                        const unsigned wps = (fl >> 8) * 128u, gps = (unsigned)pl.M * 128u;    // grad_value rows: fp32
                        const unsigned gbase = ra[i].x * (128u / ROWB);     // same pixel, fp32 rows (mod 2^32 like the offset)
                        const unsigned dg[4] = {0u, gps, wps, wps + gps};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if (fpath & (1u << k)) {
                                const float wa = wk[k] * a_s;
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(
                                        (wa * gs[j]) * ci[j] * li, gr,
                                        (int)(gbase + dg[k] + (unsigned)(cpair[j >> 1] + (j & 1)) * 4u), 0, 0);
                            }
                        }
                    }
                    // ---- grad_loc / grad_attn of this point ----
                    const f32x4 tga = g * a;
                    const f32x4 val = wk[0] * v[i][0] + wk[1] * v[i][1] + wk[2] * v[i][2] + wk[3] * v[i][3];
                    const f32x4 gw = hh * (v[i][1] - v[i][0]) + lh * (v[i][3] - v[i][2]);
                    const f32x4 gh = hw * (v[i][2] - v[i][0]) + lw * (v[i][3] - v[i][1]);
                    float pa = g.x * val.x + g.y * val.y + g.z * val.z + g.w * val.w;
                    float pw = gw.x * tga.x + gw.y * tga.y + gw.z * tga.z + gw.w * tga.w;
                    float ph = gh.x * tga.x + gh.y * tga.y + gh.z * tga.z + gh.w * tga.w;
                    pa = sum8(pa);
                    pw = sum8(pw);
                    ph = sum8(ph);
                    if (sub == (t & 7)) {     // results recycle words 1-3 of the point's record
                        f32_alias *slot = reinterpret_cast<f32_alias *>(&rec[2 * t]);
                        slot[1] = pw;
                        slot[2] = ph;
                        slot[3] = pa;
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- the row's grad_loc / grad_attn (or their images under the prologue's Jacobians) ----
        if (row.ok) {
            const f32_alias *res = reinterpret_cast<const f32_alias *>(rec);   // point t: res[8 t + 1..3], a: res[8 t + 6]
            if (FUSED) {
                // softmax Jacobian with the TRUE weights (the records carry 0 for gated-off points, whose logits still
                // receive -a_t * sum_j a_j grad_attn_j)
                float dot = 0.f;
                for (int t = sub; t < LP; t += 8) dot += (expf(lg[t] - mx) * rsum) * res[8 * t + 3];
                dot = row_sum<8>(dot);
                float *gp = grad_proj + qrow * src.proj_stride;
                for (int t = sub; t < LP; t += 8)
                    gp[src.n_off + m * LP + t] = (expf(lg[t] - mx) * rsum) * (res[8 * t + 3] - dot);
                for (int i = sub; i < 2 * LP; i += 8) {
                    const int t = i >> 1, comp = i & 1, l = t / P;
                    const float size = (float)(comp ? tb.H[l] : tb.W[l]);
                    const float gl = res[8 * t + 1 + comp] * size;
                    float go;
                    if (src.ref_dim == 2) {
                        go = gl / size;
                    } else {
                        const float *rp = src.ref + (qrow * L + l) * 4;
                        go = gl * (rp[2 + comp] * (0.5f / (float)P));
                    }
                    gp[m * 2 * LP + i] = go;
                }
            } else {
                for (int i = sub; i < 2 * LP; i += 8) {
                    const int t = i >> 1, comp = i & 1, l = t / P;
                    grad_loc[row.pm * LP * 2 + i] = res[8 * t + 1 + comp] * (float)(comp ? tb.H[l] : tb.W[l]);
                }
                for (int t = sub; t < LP; t += 8) grad_attn[row.pm * LP + t] = res[8 * t + 3];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // ---- flush: one coalesced global float atomic per touched window element ----
    // (32 consecutive lanes cover one 128-byte row: the fastest pattern of the L2 atomic units)
    const unsigned long long *win_u64 = reinterpret_cast<const unsigned long long *>(s_dyn);
    for (int l = 0; l < ((pl.ablate & 1) ? 0 : L); ++l) {
        const int win = tb.win[l], magic = tb.magic[l], base = tb.base[l];
        const int oy = tb.oy[l], ox = tb.ox[l], H = tb.H[l], W = tb.W[l];
        const float li = s_linv[l];
        const int n = win * win * D;
        for (int i = threadIdx.x; i < n; i += kTileThreads) {
            const int pix = i >> 5, c = i & 31;
            const unsigned long long tot = win_u64[(base + pix) * (D / 2) + (c >> 1)];
            const int lo = (int)(unsigned)(tot & 0xffffffffull);
            const int q = (c & 1) ? (int)(unsigned)(tot >> 32) + (lo < 0 ? 1 : 0) : lo;
            if (q != 0) {
                const int wy = (pix * magic) >> 16, wx = pix - wy * win;
                const int gy = oy + wy, gx = ox + wx;
                if (gy < H && gx < W)
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(
                        (float)q * s_cinv[c] * li, gr,
                        (int)(tile_pixel_off<128u>(tb, pl, b, l, gy, gx, m) + (unsigned)c * 4u), 0, 0);
            }
        }
    }
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
            pa[p] = FUSED ? expf(lg[t] - mx) * rsum : src.attn[pm * (unsigned)LP + (unsigned)t];
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

// softmax Jacobian of the fused backward, in place on the logit columns of grad_proj (they hold d/d attention):
// grad_logit_t = a_t (ga_t - sum_j a_j ga_j); 8 lanes per (query, head) row.
__global__ __launch_bounds__(256) void msda_softmax_jacobian_kernel(const PointSrc fs, long n_rows, int M, int LP,
                                                                    float *__restrict__ grad_proj) {
    const int sub = threadIdx.x & 7;
    for (long pm0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 3; pm0 < ((n_rows + 7) & ~7L);
         pm0 += ((long)gridDim.x * blockDim.x) >> 3) {
        const bool ok = pm0 < n_rows;
        const long pm = ok ? pm0 : n_rows - 1;
        const long qrow = pm / M;
        const int m = (int)(pm - qrow * M);
        const float *lg = fused_logits(fs, qrow, m, LP);
        float *ga = grad_proj + qrow * fs.proj_stride + fs.n_off + (long)m * LP;
        float mx, rsum;
        row_softmax_stats<8>(lg, LP, sub, mx, rsum);
        float dot = 0.f;
        for (int t = sub; t < LP; t += 8) dot += (expf(lg[t] - mx) * rsum) * ga[t];
        dot = row_sum<8>(dot);
        if (ok)
            for (int t = sub; t < LP; t += 8) ga[t] = (expf(lg[t] - mx) * rsum) * (ga[t] - dot);
    }
}

// Split fused backward, last step.  The plain tiled kernel has left d/d(sampling location) in the offset columns and
// d/d(attention) in the logit columns of grad_proj; `fs.attn` is the workspace copy of the softmax weights the
// prologue kernel wrote.  In place: offsets <- location Jacobian (ms_deform_attn.py:114-120 of the reference module),
// logits <- softmax Jacobian  a_t (ga_t - sum_j a_j ga_j).  8 lanes per (query, head) row.
__global__ __launch_bounds__(256) void msda_fused_finish_kernel(const int64_t *__restrict__ shapes, const PointSrc fs,
                                                               long n_rows, int M, int L, int P,
                                                               float *__restrict__ grad_proj) {
    const int LP = L * P;
    const int sub = threadIdx.x & 7;
    for (long pm0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 3; pm0 < ((n_rows + 7) & ~7L);
         pm0 += ((long)gridDim.x * blockDim.x) >> 3) {
        const bool ok = pm0 < n_rows;
        const long pm = ok ? pm0 : n_rows - 1;
        const long qrow = pm / M;
        const int m = (int)(pm - qrow * M);
        const float *a = fs.attn + pm * LP;
        float *ga = grad_proj + qrow * fs.proj_stride + fs.n_off + (long)m * LP;
        float *gl = grad_proj + qrow * fs.proj_stride + (long)m * LP * 2;
        float dot = 0.f;
        for (int t = sub; t < LP; t += 8) dot += a[t] * ga[t];
        dot = row_sum<8>(dot);
        if (!ok) continue;
        for (int t = sub; t < LP; t += 8) ga[t] = a[t] * (ga[t] - dot);
        for (int i = sub; i < 2 * LP; i += 8) {
            const int t = i >> 1, comp = i & 1, l = t / P;
            const float g = gl[i];
            if (fs.ref_dim == 2) {
                gl[i] = g / (float)shapes[2 * l + 1 - comp];                    // x / W_l, y / H_l
            } else {
                const float *rp = fs.ref + (qrow * L + l) * 4;
                gl[i] = g * (rp[2 + comp] * (0.5f / (float)P));
            }
        }
    }
}

// The same for L*P <= 16 with one lane per point: coalesced reads / writes of the three column groups.
__global__ __launch_bounds__(256) void msda_fused_finish16_kernel(const int64_t *__restrict__ shapes, const PointSrc fs,
                                                                 long n_rows, int M, int L, int P,
                                                                 float *__restrict__ grad_proj, int offsets_done) {
    const int LP = L * P;
    const int t = threadIdx.x & 15;
    const long rows_pad = (n_rows + 3) & ~3L;
    for (long pm0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4; pm0 < rows_pad;
         pm0 += ((long)gridDim.x * blockDim.x) >> 4) {
        const bool ok = pm0 < n_rows && t < LP;
        const long pm = pm0 < n_rows ? pm0 : n_rows - 1;
        const long qrow = pm / M;
        const int m = (int)(pm - qrow * M);
        float *ga = grad_proj + qrow * fs.proj_stride + fs.n_off + (long)m * LP;
        float *gl = grad_proj + qrow * fs.proj_stride + (long)m * LP * 2;
        const float a = t < LP ? fs.attn[pm * LP + t] : 0.f;
        const float g = t < LP ? ga[t] : 0.f;
        float dot = a * g;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 16);
        if (!ok) continue;
        ga[t] = a * (g - dot);
        if (offsets_done) continue;       // (the producer wrote the final offset gradients: 2-d reference points)
        const int l = t / P;
        f32x2 d = *reinterpret_cast<f32x2 *>(gl + 2 * t);
        if (fs.ref_dim == 2) {
            d.x = d.x / (float)shapes[2 * l + 1];
            d.y = d.y / (float)shapes[2 * l];
        } else {
            const float *rp = fs.ref + (qrow * L + l) * 4;
            d.x = d.x * (rp[2] * (0.5f / (float)P));
            d.y = d.y * (rp[3] * (0.5f / (float)P));
        }
        *reinterpret_cast<f32x2 *>(gl + 2 * t) = d;
    }
}

// The two side kernels of the slim split backward with ONE lane per (query, head) row (L*P == 16, row pitches multiples
// of 4): the sixteen logits / weights / gradients of a row are four 16-byte accesses of that lane, the softmax and its
// Jacobian run in registers -- no cross-lane traffic, a sixteenth of the threads, one index division per row.  The sums
// associate exactly like the 16-lane butterflies above ((t, t + 8) first, then 4, 2, 1 / 1, 2, 4), so the bits are theirs.
__global__ __launch_bounds__(256) void msda_fused_attn16_rows_kernel(const PointSrc fs, unsigned n_rows, unsigned M,
                                                                    float *__restrict__ attn_out) {
    for (unsigned pm = blockIdx.x * blockDim.x + threadIdx.x; pm < n_rows; pm += gridDim.x * blockDim.x) {
        const unsigned qrow = pm / M, m = pm - qrow * M;
        const f32x4 *lp = reinterpret_cast<const f32x4 *>(fs.proj + ((size_t)qrow * (unsigned)fs.proj_stride +
                                                                   (unsigned)fs.n_off + m * 16u));
        float lg[16];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4 v = lp[k];
            lg[4 * k] = v.x; lg[4 * k + 1] = v.y; lg[4 * k + 2] = v.z; lg[4 * k + 3] = v.w;
        }
        float mx = lg[0];
#pragma unroll
        for (int t = 1; t < 16; ++t) mx = fmaxf(mx, lg[t]);
        float e[16], s8[8];
#pragma unroll
        for (int t = 0; t < 16; ++t) e[t] = expf(lg[t] - mx);
#pragma unroll
        for (int t = 0; t < 8; ++t) s8[t] = e[t] + e[t + 8];
        const float sum = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
        const float rsum = 1.f / sum;
        f32x4 *op = reinterpret_cast<f32x4 *>(attn_out + (size_t)pm * 16u);
#pragma unroll
        for (int k = 0; k < 4; ++k) op[k] = f32x4{e[4 * k] * rsum, e[4 * k + 1] * rsum, e[4 * k + 2] * rsum, e[4 * k + 3] * rsum};
    }
}

__global__ __launch_bounds__(256) void msda_fused_finish16_rows_kernel(const PointSrc fs, unsigned n_rows, unsigned M,
                                                                      float *__restrict__ grad_proj) {
    for (unsigned pm = blockIdx.x * blockDim.x + threadIdx.x; pm < n_rows; pm += gridDim.x * blockDim.x) {
        const unsigned qrow = pm / M, m = pm - qrow * M;
        f32x4 *gp = reinterpret_cast<f32x4 *>(grad_proj + ((size_t)qrow * (unsigned)fs.proj_stride + (unsigned)fs.n_off + m * 16u));
        const f32x4 *ap = reinterpret_cast<const f32x4 *>(fs.attn + (size_t)pm * 16u);
        float a[16], g[16], d[16];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4 av = ap[k], gv = gp[k];
            a[4 * k] = av.x; a[4 * k + 1] = av.y; a[4 * k + 2] = av.z; a[4 * k + 3] = av.w;
            g[4 * k] = gv.x; g[4 * k + 1] = gv.y; g[4 * k + 2] = gv.z; g[4 * k + 3] = gv.w;
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) d[t] = a[t] * g[t];
#pragma unroll
        for (int t = 0; t < 8; ++t) d[t] += d[t + 8];
#pragma unroll
        for (int t = 0; t < 4; ++t) d[t] += d[t + 4];
        const float dot = (d[0] + d[2]) + (d[1] + d[3]);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            gp[k] = f32x4{a[4 * k] * (g[4 * k] - dot), a[4 * k + 1] * (g[4 * k + 1] - dot),
                          a[4 * k + 2] * (g[4 * k + 2] - dot), a[4 * k + 3] * (g[4 * k + 3] - dot)};
    }
}

#include "msda_fwd_win.h"
#include "msda_bwd_rows.h"
#include "msda_bwd_bins.h"

// ----------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------
thread_local char g_err[256] = {0};
thread_local const char *g_kernel = "";

std::atomic<int> opt_fwd_variant{0}, opt_bwd_variant{0};
std::atomic<int> opt_fwd_block{256}, opt_bwd_block{256};
std::atomic<int> opt_fwd_grid_mult{32}, opt_bwd_grid_mult{16};
// (backward margin 4, round 3: +1 % at the initialisation's offsets, -21 / -37 % when they are 1.5x / 2x larger,
//  profiles/r03_bwd_margin_sweep.txt)
std::atomic<int> opt_fwd_tile_margin{3}, opt_bwd_tile_margin{4};
std::atomic<int> opt_fwd_tile_l0{1};      // hybrid forward: first level served from LDS windows
std::atomic<int> opt_bwd_split{1};        // fused backward with a workspace: prologue kernel + plain tiled kernel + finish kernel
std::atomic<int> opt_bwd_wide_log2{12};   // tiled backward: row-magnitude range (log2) that makes a region "wide"; 0 = off
std::atomic<int> opt_bwd_ablate{0};
std::atomic<int> opt_bwd_rows_block{0};   // threads per workgroup of msda_bwd_d32_rows (0: by problem size)
std::atomic<int> opt_bwd_bins_strip{4};   // counting-sort backward: region rows per strip of the block -> region walk
std::atomic<int> opt_bwd_bins_margin{6};     // small-margin level (level 0 of the selector)
std::atomic<int> opt_bwd_bins_margin_hi{9};  // large-margin level (level 1): the largest window that keeps 5 workgroups per CU
std::atomic<int> opt_auto_select{1};         // msda_select.h: follow the measured off-window share (0: level 0 always)
std::atomic<int> opt_sel_level{-1};          // >= 0: pin the selector's level (tests, benchmarks)
std::atomic<int> opt_sel_up0{5}, opt_sel_up1{100}, opt_sel_down1{2}, opt_sel_down2{60};   // backward thresholds, 1/1000 of the valid corners
std::atomic<int> opt_sel_fwd_up{50}, opt_sel_fwd_down{20};                                  // forward thresholds  // counting-sort backward: window margin (the window is only a table of counters)
std::atomic<int> opt_bwd_rows{1};         // 0: few-query D = 32 calls keep the generic row-per-block backward
std::atomic<int> opt_fwd_head_major{0};    // gather forward: head-major task walk (one head per XCD)
std::atomic<int> opt_fwd_win_rlog{3};       // windowed forward: log2 of the region height on level 0
std::atomic<int> opt_fwd_win_rlogx{3};      // log2 of the region width (at least the height)
std::atomic<int> opt_fwd_win_auto{1};       // 0: never pick the windowed forward on its own
std::atomic<int> opt_fwd_win_block{256};    // threads per workgroup (256 / 512)
std::atomic<int> opt_fwd_win_l0{1};         // first level served from an LDS window
std::atomic<int> opt_fwd_win_margins{0x3333};  // window margin per level, 4 bits each (level 0 in the low nibble)
std::atomic<int> opt_fwd_win_ablate{0};     // profiling only
std::atomic<int> opt_fwd_win_wps{0};        // 3 / 4: force the 168- / 128-register build
std::atomic<int> opt_fwd_win_early{2};      // 0 / 1: global points after / around the LDS phase (2: by register budget)
std::atomic<int> opt_bwd_side_rows{1};      // slim split backward: side kernels with one lane per (query, head) row (0: one lane per point)
std::atomic<int> opt_fwd_win_trace_lo{0}, opt_fwd_win_trace_hi{0};   // profiling: device address of the timeline buffer (31 + 31 bits)
std::atomic<int> opt_fwd_win_dma{1};        // fill the windows with buffer_load ... lds       // profiling only: drop parts of the tiled backward (results are then wrong)

int fail(int code, const char *msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

int check_launch(const char *what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    g_err[0] = 0;
    return MSDA_OK;
}

int check_dims(const void *a, const void *b, const void *c, const void *d, const void *e, const void *f, int N, int S,
               int M, int D, int L, int Lq, int P) {
    if (!a || !b || !c || !d || !e || !f) return fail(MSDA_EINVAL, "null pointer argument");
    if (N < 0 || Lq < 0) return fail(MSDA_EINVAL, "negative batch/query count");
    if (S <= 0 || M <= 0 || D <= 0 || L <= 0 || P <= 0) return fail(MSDA_EINVAL, "non-positive dimension");
    // the reference kernels index with 32-bit ints (.cuh:255-270); keep the same envelope, loudly
    const double lim = 2147483647.0;
    if ((double)N * S * M * D > lim || (double)N * Lq * M * L * P * 2 > lim || (double)N * Lq * M * D > lim)
        return fail(MSDA_ERANGE, "tensor exceeds 2^31 elements");
    return MSDA_OK;
}

int clamp_grid(long want, int mult) {
    long cap = (long)kNumCU * (mult > 0 ? mult : 8);
    if (want < 1) want = 1;
    return (int)(want < cap ? want : cap);
}

// the specialised kernels address `value` (and the fp32 grad_value) with 32-bit byte offsets
bool d32_ok(int D, int L, long value_elems) { return D == 32 && L <= kMaxLevels && value_elems * 4 < 0x7fffff00L; }

// Plan the region tiling from the HOST copy of the level shapes.  Returns false when the tiled
// kernels do not apply (then the gather / generic kernels run).  Levels below `l0` get no window.
bool make_tile_plan(TilePlan &pl, const int64_t *shapes_host, int N, int S, int M, int D, int L, int Lq, int P,
                    long value_bytes, int margin, int l0, size_t lds_rows_extra, size_t fixed_lds, size_t &lds,
                    bool check_lds = true) {
    if (!shapes_host || D != 32 || L < 1 || L > kTileMaxL || Lq != S || L * P > kMaxFusedLP) return false;
    if (margin < 0) margin = 0;
    if (l0 < 0) l0 = 0;
    if (l0 > L) l0 = L;
    memset(&pl, 0, sizeof(pl));
    pl.N = N; pl.S = S; pl.M = M; pl.L = L; pl.P = P; pl.Lq = Lq; pl.l0 = l0;
    pl.value_bytes = (unsigned)value_bytes;
    long q = 0;
    int rows = 0, px = 0, RY = 0, RX = 0;
    for (int l = 0; l < kTileMaxL; ++l) {
        if (l < L) {
            const long H = shapes_host[2 * l], W = shapes_host[2 * l + 1];
            if (H <= 0 || W <= 0 || H > 32767 || W > 32767 || W * M >= (1L << 23)) return false;
            const int sh = L - 1 - l, side = 1 << sh;
            int win = side + 2 * margin;
            if (win > 32) win = 32;
            if (l < l0) win = 0;
            pl.H[l] = (int)H; pl.W[l] = (int)W; pl.qstart[l] = (int)q; pl.shift[l] = sh; pl.row0[l] = rows;
            pl.win[l] = win; pl.win_base[l] = px;
            int magic = 65537;
            if (win > 0) {
                magic = 65536 / win + 1;
                for (int x = 0; x < win * win; ++x)
                    if (((x * magic) >> 16) != x / win) return false;
            }
            pl.win_magic[l] = magic;
            q += H * W; rows += side * side; px += win * win;
            const int ry = (int)((H + side - 1) / side), rx = (int)((W + side - 1) / side);
            RY = ry > RY ? ry : RY;
            RX = rx > RX ? rx : RX;
        } else {  // inert padding so that table loads stay in range
            pl.H[l] = 1; pl.W[l] = 1; pl.qstart[l] = (int)q; pl.shift[l] = 0; pl.row0[l] = rows;
            pl.win[l] = 0; pl.win_magic[l] = 65537; pl.win_base[l] = px;
        }
    }
    if (q != S) return false;  // host shapes do not describe this value tensor
    if (px + (int)lds_rows_extra >= 65535) return false;    // window cells travel as 16-bit indices
    pl.row0[kTileMaxL] = rows; pl.win_base[kTileMaxL] = px;
    for (int l = L; l <= kTileMaxL; ++l) { pl.row0[l] = rows; pl.win_base[l] = px; }
    pl.rows = rows; pl.RY = RY; pl.RX = RX;
    if (rows > kTileMaxRows) return false;
    const long nb = (long)N * RY * RX * M;
    if (nb > (1L << 30)) return false;
    pl.n_blocks = (int)nb;
    lds = ((size_t)px + lds_rows_extra) * 128 + fixed_lds;
    return !check_lds || lds <= 160 * 1024 - 2048;
}

// LDS layout of msda_bwd_d32_bins (msda_bwd_bins.h) for `ni` items per thread; false when the call does not fit.
bool make_bins_plan(BinsPlan &bp, const TilePlan &pl, int ni, size_t &lds) {
    const int P = pl.P, n_items = pl.rows * P;
    if (P < 1 || n_items < 1 || n_items > kTileThreads * ni) return false;
    const int magic = 65536 / P + 1;
    for (int i = 0; i < kTileThreads * ni; ++i)
        if (((i * magic) >> 16) != i / P) return false;
    int win_max = 0;
    for (int l = 0; l < pl.L; ++l) win_max = pl.win[l] > win_max ? pl.win[l] : win_max;
    const int ncell = win_max * win_max;
    const int C = (((ncell + 63) / 64) + 3) & ~3;
    auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    size_t o = up16((size_t)(pl.rows + 1) * kBinsGRow);
    bp.n_items = n_items;
    bp.magic_p = magic;
    bp.scan_c = C;
    bp.strip = opt_bwd_bins_strip.load() > 0 ? opt_bwd_bins_strip.load() : 1;
    bp.o_x = (unsigned)o;                       // records + flags, later the sorted entries
    bp.o_fl = (unsigned)(o + (size_t)n_items * 16);
    const size_t rec_bytes = up16((size_t)n_items * 20), ent_bytes = up16((size_t)(4 * n_items + 1) * 8);
    o += rec_bytes > ent_bytes ? rec_bytes : ent_bytes;
    // union: ticket counters + row tables (dead after the sort) | flush transpose
    bp.o_st = (unsigned)o;
    bp.o_cnt = (unsigned)o;
    bp.o_rowp = (unsigned)(o + (size_t)C * 64 * 4);
    bp.o_rowa = (unsigned)(bp.o_rowp + up16((size_t)(pl.rows + 1) * 4));
    bp.o_rowq = (unsigned)(bp.o_rowa + up16((size_t)(pl.rows + 1) * 4));
    const size_t tab_bytes = (size_t)C * 64 * 4 + 3 * up16((size_t)(pl.rows + 1) * 4);
    const size_t st_bytes = (size_t)(kTileThreads / 64) * kBinsStageWave;
    o += tab_bytes > st_bytes ? tab_bytes : st_bytes;
    bp.o_start = (unsigned)o;
    o += up16((size_t)C * 64 * 2);
    bp.o_comp = (unsigned)o;
    o += (size_t)C * 64 * 4;
    bp.o_misc = (unsigned)o;
    o += 16 + (size_t)(kTileThreads / 64) * 16 * 4 + 16;
    lds = o;
    return lds <= 64 * 1024;
}

// Dynamic LDS above 64 KiB needs an opt-in per kernel; do it once per kernel and device for the full
// 160 KiB (the call costs host time, too much to repeat per launch).
template <typename K>
int allow_big_lds(K kernel, size_t lds) {
    if (lds <= 64 * 1024) return MSDA_OK;
    static std::atomic<unsigned long long> done{0};   // one bit per device ordinal (per template instance)
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return MSDA_OK;
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    done.fetch_or(bit, std::memory_order_release);
    return MSDA_OK;
}

// ---- msda_select.h, host side: the records and their table ----
SelSlot g_sel[kSelSlots];
std::mutex g_sel_mu;
thread_local unsigned long long g_site = 0;
thread_local int g_sel_level = 0;           // level of this thread's last selected call (msda_selector_last)
thread_local float g_sel_frac = -1.f, g_sel_inner = -1.f;

// The record of (call site, geometry); created on first use unless `stream` is capturing (allocation is not
// capturable) -- then, and when the table is full, null: the call runs at level 0 without statistics.
SelSlot *sel_acquire(int kind, int N, int S, int M, int L, int P, int Lq, int dt, hipStream_t stream) {
    if (!opt_auto_select.load()) return nullptr;
    SelKey k;
    memset(&k, 0, sizeof(k));
    (void)hipGetDevice(&k.dev);
    k.kind = kind; k.site = g_site; k.N = N; k.S = S; k.M = M; k.L = L; k.P = P; k.Lq = Lq; k.dt = dt;
    std::lock_guard<std::mutex> lock(g_sel_mu);
    SelSlot *free_slot = nullptr;
    for (int i = 0; i < kSelSlots; ++i) {
        if (g_sel[i].used && g_sel[i].key == k) return &g_sel[i];
        if (!g_sel[i].used && !free_slot) free_slot = &g_sel[i];
    }
    if (!free_slot) return nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return nullptr;
    }
    unsigned *dev = nullptr, *host = nullptr, *host_dev = nullptr;
    if (hipMalloc((void **)&dev, kSelDevWords * 4) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipHostMalloc((void **)&host, 32, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&host_dev, host, 0) != hipSuccess ||
        hipMemset(dev, 0, kSelDevWords * 4) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(dev);
        if (host) (void)hipHostFree(host);
        return nullptr;
    }
    for (int i = 0; i < 8; ++i) host[i] = 0u;
    SelSlot &s = *free_slot;
    s.key = k; s.dev = dev; s.host = host; s.host_dev = host_dev; s.seen = 0u; s.level = 0; s.calls = 0u; s.launches = 0u;
    s.frac = s.frac_inner = 0.f;
    s.used = true;
    return &s;
}

// Read what the last completed launch left in the host record, move the level, return the level for THIS call
// (a level without windows probes one level down every kSelProbeEvery-th call: `probe` is then set).
int sel_level(SelSlot *s, int kind, bool &probe) {
    probe = false;
    const int pinned = opt_sel_level.load();
    if (!s) return pinned >= 0 ? pinned : 0;
    std::lock_guard<std::mutex> lock(g_sel_mu);
    const unsigned seq = s->host[3];
    if (seq != s->seen) {
        std::atomic_thread_fence(std::memory_order_acquire);
        const unsigned valid = s->host[0], off = s->host[1], inner = s->host[2];
        s->seen = seq;
        if (valid > 0u) {
            s->frac = (float)off / (float)valid;
            s->frac_inner = (float)inner / (float)valid;
            SelRule r;
            if (kind == 0) { r.up0 = opt_sel_fwd_up.load(); r.down1 = opt_sel_fwd_down.load(); r.up1 = r.down2 = 0; }
            else { r.up0 = opt_sel_up0.load(); r.up1 = opt_sel_up1.load(); r.down1 = opt_sel_down1.load(); r.down2 = opt_sel_down2.load(); }
            // the launch stored the level it ran at: a probe ran one level below the top (and is judged by the top
            // level's rule); statistics of a launch from before the last move are dropped
            const int top = kind == 0 ? 1 : 2;
            const int ran = (int)s->host[4];
            if (s->level == top && ran == top - 1) s->level = sel_next_level(kind, top, s->frac * 1000.f, s->frac_inner * 1000.f, r);
            else if (ran == s->level) s->level = sel_next_level(kind, ran, s->frac * 1000.f, s->frac_inner * 1000.f, r);
        }
    }
    ++s->calls;
    g_sel_frac = s->seen ? s->frac : -1.f;
    g_sel_inner = s->seen ? s->frac_inner : -1.f;
    int level = pinned >= 0 ? pinned : s->level;
    const int top = kind == 0 ? 1 : 2;
    if (pinned < 0 && level == top && (s->calls % kSelProbeEvery) == 0u) {
        level = top - 1;
        probe = true;
    }
    g_sel_level = level;
    return level;
}

struct FusedArgs {     // null proj = the plain operator
    const float *proj = nullptr;
    int proj_stride = 0;
    const float *ref = nullptr;
    int ref_dim = 0;
    const unsigned char *mask = nullptr;
};

PointSrc make_src(const void *loc, const void *attn, const FusedArgs &fa, int M, int L, int P) {
    PointSrc s;
    s.loc = (const float *)loc;
    s.attn = (const float *)attn;
    s.proj = fa.proj;
    s.ref = fa.ref;
    s.mask = fa.mask;
    s.proj_stride = fa.proj_stride;
    s.n_off = 2 * M * L * P;
    s.ref_dim = fa.ref_dim;
    return s;
}

int check_fused(const FusedArgs &fa, int M, int L, int P) {
    if (!fa.proj || !fa.ref) return fail(MSDA_EINVAL, "null pointer argument");
    if (fa.ref_dim != 2 && fa.ref_dim != 4) return fail(MSDA_EINVAL, "reference points must have 2 or 4 coordinates");
    if (fa.proj_stride < 3 * M * L * P) return fail(MSDA_EINVAL, "projection rows shorter than 3*M*L*P");
    if (L * P > kMaxFusedLP) return fail(MSDA_ENOTSUP, "fused prologue supports at most 64 sampling points per head");
    return MSDA_OK;
}

template <typename TV, typename TC>
int forward_impl(const TV *value, const int64_t *shapes, const int64_t *lstart, const TC *loc, const TC *attn,
                 const FusedArgs &fa, int N, int S, int M, int D, int L, int Lq, int P, TV *out,
                 const int64_t *shapes_host, hipStream_t stream) {
    const bool fused = fa.proj != nullptr;
    int rc = fused ? check_dims(value, shapes, lstart, fa.proj, fa.ref, out, N, S, M, D, L, Lq, P)
                   : check_dims(value, shapes, lstart, loc, attn, out, N, S, M, D, L, Lq, P);
    if (rc) return rc;
    if (fused && (rc = check_fused(fa, M, L, P))) return rc;
    if ((long)N * Lq == 0) { g_err[0] = 0; return MSDA_OK; }
    int variant = opt_fwd_variant.load();
    const long value_elems = (long)N * S * M * D;
    const long value_bytes = value_elems * (long)sizeof(TV);
    constexpr bool kD32Type = sizeof(TC) == 4 && (sizeof(TV) == 4 || sizeof(TV) == 2);
    const bool can32 = kD32Type && d32_ok(D, L, value_elems);
    SelSlot *slot = nullptr;
    int sel = 0;
    bool sel_head_major = false;
    if (variant == 0) {
        // self-attention over the pyramid (one query per pixel): coarse levels from per-head LDS windows; every
        // other D = 32 call: direct gather with 4 points (16 rows) in flight -- best of the sweeps in profiles/
        const bool pyramid = can32 && sizeof(TV) == 4 && shapes_host != nullptr && Lq == S && L <= kWinMaxL &&
                             L * P <= 16 && opt_fwd_win_auto.load() != 0;
        variant = pyramid ? 12 : (can32 ? 3 : 1);
        if (pyramid) {      // msda_select.h: windows while the points stay near their queries, else the head-major gather
            slot = sel_acquire(0, N, S, M, L, P, Lq, (int)sizeof(TV), stream);
            bool probe = false;
            sel = sel_level(slot, 0, probe);
            if (sel >= 1) { variant = 3; sel_head_major = true; }
        }
    } else if (variant == 12 && can32 && shapes_host != nullptr) {
        slot = sel_acquire(0, N, S, M, L, P, Lq, (int)sizeof(TV), stream);     // forced: the selector only measures
        bool probe = false;
        (void)sel_level(slot, 0, probe);
        sel = 0;
    }
    if (variant >= 2 && !can32) variant = 1;
    if (variant == 1) {
        const long total = (long)N * Lq * M * D;
        const int grid = clamp_grid((total + 255) / 256, 32);
        if constexpr (sizeof(TC) == 4) {
            const PointSrc src = make_src(loc, attn, fa, M, L, P);
            if (fused) {
                g_kernel = "msda_fwd_generic<fused>";
                hipLaunchKernelGGL((msda_fwd_generic<TV, TC, true>), dim3(grid), dim3(256), 0, stream, value, shapes,
                                   lstart, (const TC *)nullptr, (const TC *)nullptr, src, N, S, M, D, L, Lq, P, out);
                return check_launch(g_kernel);
            }
        }
        g_kernel = "msda_fwd_generic";
        hipLaunchKernelGGL((msda_fwd_generic<TV, TC, false>), dim3(grid), dim3(256), 0, stream, value, shapes, lstart,
                           loc, attn, PointSrc{}, N, S, M, D, L, Lq, P, out);
        return check_launch(g_kernel);
    }
    if constexpr (kD32Type) {
        const PointSrc src = make_src(loc, attn, fa, M, L, P);
        if constexpr (sizeof(TV) == 4) {
            if (variant == 12) {
                WinPlan wp;
                size_t lds = 0;
                const int mgs = opt_fwd_win_margins.load();
                const int margins[kWinMaxL] = {mgs & 15, (mgs >> 4) & 15, (mgs >> 8) & 15, (mgs >> 12) & 15};
                int threads = opt_fwd_win_block.load();
                if (threads != 512 && threads != 384 && threads != 128) threads = 256;
                if (make_win_plan(wp, shapes_host, N, S, M, D, L, Lq, P, value_bytes, opt_fwd_win_rlogx.load(),
                                  opt_fwd_win_rlog.load(), opt_fwd_win_l0.load(), margins, threads, lds) &&
                    !(src.mask != nullptr && wp.wgroups_max > 8 * (threads / 64))) {
                    const int grid = (wp.n_blocks + 7) & ~7;
                    const bool dma = opt_fwd_win_dma.load() != 0;
#define MSDA_LAUNCH_WIN(FU, DM, WPS, EA, NAME)                                                                       \
    do {                                                                                                             \
        rc = allow_big_lds(msda_fwd_d32_win<FU, DM, WPS, EA>, lds);                                                  \
        if (rc) return rc;                                                                                           \
        g_kernel = NAME;                                                                                             \
        hipLaunchKernelGGL((msda_fwd_d32_win<FU, DM, WPS, EA>), dim3(grid), dim3(threads), lds, stream,              \
                           (const float *)value, lstart, src, (float *)out, wp);                                     \
    } while (0)
#define MSDA_LAUNCH_WIN_T(FU, NAME)                                                                                  \
    do {                                                                                                             \
        rc = allow_big_lds(msda_fwd_d32_win<FU, true, 3, true, true>, lds);                                          \
        if (rc) return rc;                                                                                           \
        g_kernel = NAME;                                                                                             \
        hipLaunchKernelGGL((msda_fwd_d32_win<FU, true, 3, true, true>), dim3(grid), dim3(threads), lds, stream,      \
                           (const float *)value, lstart, src, (float *)out, wp);                                     \
    } while (0)
                    wp.ablate = opt_fwd_win_ablate.load();
                    wp.trace = reinterpret_cast<unsigned long long *>(((unsigned long long)opt_fwd_win_trace_hi.load() << 31) |
                                                                      (unsigned long long)opt_fwd_win_trace_lo.load());
                    {       // (a captured launch would replay one parity for ever: no statistics from inside a capture)
                        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                        const bool capturing = hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
                        if (slot && !capturing) {
                            std::lock_guard<std::mutex> lock(g_sel_mu);
                            wp.stats = slot->dev;
                            wp.stats_host = slot->host_dev;
                            wp.sel_parity = (int)(slot->launches++ & 1u);
                            wp.sel_level = sel;
                        } else {
                            (void)hipGetLastError();
                        }
                    }
                    // register budget by what the LDS footprint admits: four 256-thread workgroups per CU (<= 40 KB
                    // each) -> 128 registers and the global points after the LDS phase; three -> 168 registers
                    int wps = opt_fwd_win_wps.load();
                    if (wps != 3 && wps != 4) wps = (threads <= 256 && lds + 640 > 40 * 1024) ? 3 : 4;
                    if (threads > 256) wps = 4;
                    int early = opt_fwd_win_early.load();
                    if (early > 1) early = wps == 3 ? 1 : 0;
                    if (!dma) {       // the register-staged fill is a debugging aid: one build
                        if (fused) MSDA_LAUNCH_WIN(true, false, 4, false, "msda_fwd_d32_win<fused,nodma>");
                        else MSDA_LAUNCH_WIN(false, false, 4, false, "msda_fwd_d32_win<nodma>");
                    } else if (fused) {
                        if (wps == 3 && early && (wp.trace || wp.ablate)) MSDA_LAUNCH_WIN_T(true, "msda_fwd_d32_win<fused,w3,early>");
                        else if (wps == 3 && early) MSDA_LAUNCH_WIN(true, true, 3, true, "msda_fwd_d32_win<fused,w3,early>");
                        else if (wps == 3) MSDA_LAUNCH_WIN(true, true, 3, false, "msda_fwd_d32_win<fused,w3>");
                        else if (early) MSDA_LAUNCH_WIN(true, true, 4, true, "msda_fwd_d32_win<fused,w4,early>");
                        else MSDA_LAUNCH_WIN(true, true, 4, false, "msda_fwd_d32_win<fused,w4>");
                    } else {
                        if (wps == 3 && early && (wp.trace || wp.ablate)) MSDA_LAUNCH_WIN_T(false, "msda_fwd_d32_win<w3,early>");
                        else if (wps == 3 && early) MSDA_LAUNCH_WIN(false, true, 3, true, "msda_fwd_d32_win<w3,early>");
                        else if (wps == 3) MSDA_LAUNCH_WIN(false, true, 3, false, "msda_fwd_d32_win<w3>");
                        else if (early) MSDA_LAUNCH_WIN(false, true, 4, true, "msda_fwd_d32_win<w4,early>");
                        else MSDA_LAUNCH_WIN(false, true, 4, false, "msda_fwd_d32_win<w4>");
                    }
#undef MSDA_LAUNCH_WIN
#undef MSDA_LAUNCH_WIN_T
                    return check_launch(g_kernel);
                }
                variant = 3;  // the windowed kernel does not apply to this call
            }
            if (variant == 8 || variant == 9) {
                TilePlan pl;
                size_t lds = 0;
                const size_t rec_bytes = (size_t)32 * (L * P + 1) * 16;
                if (L * P <= 16 &&
                    make_tile_plan(pl, shapes_host, N, S, M, D, L, Lq, P, value_bytes, opt_fwd_tile_margin.load(),
                                   opt_fwd_tile_l0.load(), 1, rec_bytes, lds)) {
                    const int grid = (pl.n_blocks + 7) & ~7;
#define MSDA_LAUNCH_HY(PTS, FU, NAME)                                                                                \
    do {                                                                                                             \
        rc = allow_big_lds(msda_fwd_d32_hybrid<PTS, FU>, lds);                                                       \
        if (rc) return rc;                                                                                           \
        g_kernel = NAME;                                                                                             \
        hipLaunchKernelGGL((msda_fwd_d32_hybrid<PTS, FU>), dim3(grid), dim3(kTileThreads), lds, stream,              \
                           (const float *)value, lstart, src, (float *)out, pl);                                     \
    } while (0)
                    if (variant == 8) {
                        if (fused) MSDA_LAUNCH_HY(4, true, "msda_fwd_d32_hybrid<4,fused>");
                        else MSDA_LAUNCH_HY(4, false, "msda_fwd_d32_hybrid<4>");
                    } else {
                        if (fused) MSDA_LAUNCH_HY(2, true, "msda_fwd_d32_hybrid<2,fused>");
                        else MSDA_LAUNCH_HY(2, false, "msda_fwd_d32_hybrid<2>");
                    }
#undef MSDA_LAUNCH_HY
                    return check_launch(g_kernel);
                }
                variant = 3;  // tiling does not apply to this call
            }
        }
        if (variant > 4) variant = 3;
        constexpr int ROWS = RowGeom<TV>::kRows;
        int block = opt_fwd_block.load();
        if (block < 64 || block > 256 || (block & 63)) block = 256;  // kernels carry __launch_bounds__(256)
        const int wpb = block / 64;
        const int head_major = (opt_fwd_head_major.load() != 0 || sel_head_major) && (long)N * Lq >= 4096 ? 1 : 0;
        const long n_tasks = head_major ? (((long)N * Lq + ROWS - 1) / ROWS) * M : ((long)N * Lq * M + ROWS - 1) / ROWS;
        // small problems: one wave per block so every task gets its own CU slot
        int use_block = block;
        if (n_tasks < (long)kNumCU * wpb) use_block = 64;
        const int uwpb = use_block / 64;
        int grid = clamp_grid((n_tasks + uwpb - 1) / uwpb, opt_fwd_grid_mult.load());
        grid = (grid + 7) & ~7;  // whole blocks per XCD residue
        const size_t lds = (size_t)uwpb * ROWS * (2 * L * P + 1) * 16;
#define MSDA_LAUNCH_FWD(PTS, FU, NAME)                                                                               \
    do {                                                                                                             \
        g_kernel = NAME;                                                                                             \
        hipLaunchKernelGGL((msda_fwd_d32_gather<PTS, TV, FU>), dim3(grid), dim3(use_block), lds, stream, value,      \
                           shapes, lstart, src, N, S, M, L, Lq, P, out, (unsigned)value_bytes, head_major);          \
    } while (0)
        if (fused) {
            if (variant == 3) MSDA_LAUNCH_FWD(4, true, sizeof(TV) == 2 ? "msda_fwd_d32_gather<4,bf16,fused>" : "msda_fwd_d32_gather<4,fused>");
            else if (variant == 4) MSDA_LAUNCH_FWD(1, true, sizeof(TV) == 2 ? "msda_fwd_d32_gather<1,bf16,fused>" : "msda_fwd_d32_gather<1,fused>");
            else MSDA_LAUNCH_FWD(2, true, sizeof(TV) == 2 ? "msda_fwd_d32_gather<2,bf16,fused>" : "msda_fwd_d32_gather<2,fused>");
        } else {
            if (variant == 3) MSDA_LAUNCH_FWD(4, false, sizeof(TV) == 2 ? "msda_fwd_d32_gather<4,bf16>" : "msda_fwd_d32_gather<4>");
            else if (variant == 4) MSDA_LAUNCH_FWD(1, false, sizeof(TV) == 2 ? "msda_fwd_d32_gather<1,bf16>" : "msda_fwd_d32_gather<1>");
            else MSDA_LAUNCH_FWD(2, false, sizeof(TV) == 2 ? "msda_fwd_d32_gather<2,bf16>" : "msda_fwd_d32_gather<2>");
        }
#undef MSDA_LAUNCH_FWD
        return check_launch(g_kernel);
    }
    return fail(MSDA_ENOTSUP, "no specialised forward for this dtype");
}

template <typename TV, typename TC, typename TG>
int backward_impl(const TV *value, const int64_t *shapes, const int64_t *lstart, const TC *loc, const TC *attn,
                  const FusedArgs &fa, const TV *grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                  TG *grad_value, TC *grad_loc, TC *grad_attn, float *grad_proj, float *grad_ref_part,
                  int zero_grad_value, const int64_t *shapes_host, hipStream_t stream, float *workspace = nullptr,
                  size_t workspace_bytes = 0) {
    const bool fused = fa.proj != nullptr;
    int rc = fused ? check_dims(value, shapes, lstart, fa.proj, fa.ref, grad_out, N, S, M, D, L, Lq, P)
                   : check_dims(value, shapes, lstart, loc, attn, grad_out, N, S, M, D, L, Lq, P);
    if (rc) return rc;
    if (fused) {
        if ((rc = check_fused(fa, M, L, P))) return rc;
        if (!grad_value || !grad_proj) return fail(MSDA_EINVAL, "null gradient pointer");
    } else if (!grad_value || !grad_loc || !grad_attn) {
        return fail(MSDA_EINVAL, "null gradient pointer");
    }
    if (zero_grad_value) {
        const hipError_t e = hipMemsetAsync(grad_value, 0, (size_t)N * S * M * D * sizeof(TG), stream);
        if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    }
    if ((long)N * Lq == 0) { g_err[0] = 0; return MSDA_OK; }
    int variant = opt_bwd_variant.load();
    const long value_elems = (long)N * S * M * D;
    const long value_bytes = value_elems * (long)sizeof(TV);
    constexpr bool kD32Type = sizeof(TC) == 4 && sizeof(TG) == 4 && (sizeof(TV) == 4 || sizeof(TV) == 2);
    const bool can_tile = kD32Type && d32_ok(D, L, value_elems) && shapes_host && Lq == S && L <= kTileMaxL &&
                          grad_ref_part == nullptr;
    // Measured on MI355X (profiles/): per-contribution global float atomics cap the backward at ~1.1 ms
    // for the encoder call (L2 atomic throughput; the row-per-block kernel's 32-consecutive-lane pattern
    // is the fastest of them).  Self-attention over the pyramid (Lq == S, host shapes known) therefore
    // takes the region-tiled kernel that pre-reduces grad_value in fixed-point LDS windows; every other call
    // (decoder queries) takes the row-per-block kernel.
    // Self-attention over the pyramid: the counting-sort kernel (msda_bwd_bins.h, round 4: 145 us at the encoder
    // shape; tile_lv 218, tile_q2 317), at the window margin the measured off-window share asks for -- or, when most
    // points leave even the large window (uniformly random locations), no windows at all (msda_select.h).
    SelSlot *slot = nullptr;
    int sel = 0, bins_margin = opt_bwd_bins_margin.load(), bins_shrink = 0;
    if (kD32Type && can_tile && (variant == 0 || variant == 12)) {
        slot = sel_acquire(1, N, S, M, L, P, Lq, (int)sizeof(TV), stream);
        bool probe = false;
        sel = sel_level(slot, 1, probe);
        if (variant == 12) sel = opt_sel_level.load() >= 0 ? sel : 0;         // forced: the selector only measures
        if (sel >= 1) {
            bins_shrink = opt_bwd_bins_margin_hi.load() - bins_margin;
            bins_margin = opt_bwd_bins_margin_hi.load();
            if (bins_shrink < 0) bins_shrink = 0;
        }
        if (variant == 0) variant = sel >= 2 ? 1 : 12;
    }
    if (variant == 0) variant = can_tile ? (P <= 8 ? 10 : 8) : 1;
    if (variant >= 2 && !can_tile) variant = 1;
    if constexpr (kD32Type) {
        if (variant == 12) {        // counting-sort gather (msda_bwd_bins.h); the one-kernel fused form stays with tile_lv
            const bool will_split = fa.proj != nullptr && opt_bwd_split.load() != 0 && workspace != nullptr &&
                                    workspace_bytes >= (size_t)N * Lq * M * L * P * 3 * sizeof(float);
            if (P > 8 || (fa.proj != nullptr && !will_split)) variant = 10;
        }
        if (variant == 10 || variant == 11 || variant == 12) {       // one pyramid level per workgroup
            TilePlan pl;
            size_t lds_all = 0;
            BinsPlan bp;
            memset(&bp, 0, sizeof(bp));
            int bins_ni = 0;
            size_t bins_lds = 0;
            bool planned = false;
            if (variant == 12) {
                if (make_tile_plan(pl, shapes_host, N, S, M, D, L, Lq, P, value_bytes, bins_margin, 0, 8, 0, lds_all, false)) {
                    for (int ni = 2; ni <= 3 && !bins_ni; ++ni)
                        if (make_bins_plan(bp, pl, ni, bins_lds)) bins_ni = ni;
                }
                planned = bins_ni != 0;
                if (!planned) variant = 10;
                bp.shrink = bins_shrink;
                bp.level = sel;
                // (a captured launch would replay one parity for ever: no statistics from inside a capture)
                hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                const bool capturing = hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
                if (slot && !capturing) {
                    std::lock_guard<std::mutex> lock(g_sel_mu);
                    bp.stats = slot->dev;
                    bp.stats_host = slot->host_dev;
                    bp.parity = (int)(slot->launches++ & 1u);
                } else {
                    (void)hipGetLastError();
                    bp.stats = bp.stats_host = nullptr;
                    bp.parity = 0;
                }
            }
            if (!planned)
                planned = P <= 8 && make_tile_plan(pl, shapes_host, N, S, M, D, L, Lq, P, value_bytes,
                                                   opt_bwd_tile_margin.load(), 0, 8, 0, lds_all);
            if (planned) {
                int win_max = 0;
                for (int l = 0; l < L; ++l) win_max = pl.win[l] > win_max ? pl.win[l] : win_max;
                const size_t lds = variant == 12 ? bins_lds
                                                 : (size_t)(win_max * win_max + 8) * 128 + (size_t)32 * (2 * P + 1) * 16;
                const int grid = (pl.n_blocks * L + 7) & ~7;
                PointSrc src = make_src(loc, attn, fa, M, L, P);
                pl.ablate = opt_bwd_ablate.load();
                pl.wide_log2 = opt_bwd_wide_log2.load();
                const long n_rows = (long)N * Lq * M;
                // Split fused backward (needs the caller's workspace): materialise the prologue once -- the tiled
                // kernel would otherwise redo the row softmax and the location arithmetic in each of its L
                // workgroups per region -- run the plain kernel on it, finish the Jacobians in place.
                const bool split = fused && opt_bwd_split.load() != 0 && workspace != nullptr &&
                                   workspace_bytes >= (size_t)n_rows * L * P * 3 * sizeof(float);
                // The counting-sort kernel computes the locations itself (one lane per point: the arithmetic is
                // cheap there) and, for 2-d reference points, writes the final offset gradients: the two side
                // kernels then move a third of the bytes (attention weights out, the softmax Jacobian in place).
                const bool slim = split && variant == 12 && L * P <= 16;
                const int offsets_done = slim && fa.ref_dim == 2 ? 1 : 0;
                bp.fused_loc = slim ? 1 : 0;
                bp.offsets_done = offsets_done;
                bool rows16 = false;
                if (split) {
                    float *loc_ws = workspace, *attn_ws = workspace + (size_t)n_rows * L * P * 2;
                    // one lane per row (16 points as four 16-byte accesses): the slim path's two side kernels
                    rows16 = slim && L * P == 16 && (fa.proj_stride % 4) == 0 && (src.n_off % 4) == 0 &&
                             (((uintptr_t)fa.proj | (uintptr_t)grad_proj | (uintptr_t)workspace) & 15) == 0 &&
                             n_rows < (1L << 31) && opt_bwd_side_rows.load() != 0;
                    if (rows16) {
                        hipLaunchKernelGGL(msda_fused_attn16_rows_kernel, dim3(clamp_grid((n_rows + 255) / 256, 32)),
                                           dim3(256), 0, stream, src, (unsigned)n_rows, (unsigned)M, attn_ws);
                    } else if (L * P <= 16) {
                        hipLaunchKernelGGL(msda_fused_points16_kernel, dim3(clamp_grid((n_rows * 16 + 255) / 256, 32)),
                                           dim3(256), 0, stream, shapes, src, n_rows, M, L, P,
                                           slim ? (float *)nullptr : loc_ws, attn_ws);
                    } else {
                        hipLaunchKernelGGL(msda_fused_points_kernel, dim3(clamp_grid((n_rows * 8 + 255) / 256, 16)),
                                           dim3(256), 0, stream, shapes, src, n_rows, M, L, P, loc_ws, attn_ws);
                    }
                    if ((rc = check_launch("msda_fused_points_kernel"))) return rc;
                    src.loc = loc_ws;
                    src.attn = attn_ws;
                }
#define MSDA_LAUNCH_LV(PTS, FU, NAME)                                                                                \
    do {                                                                                                             \
        rc = allow_big_lds(msda_bwd_d32_tile_lv<PTS, TV, FU>, lds);                                                  \
        if (rc) return rc;                                                                                           \
        g_kernel = NAME;                                                                                             \
        hipLaunchKernelGGL((msda_bwd_d32_tile_lv<PTS, TV, FU>), dim3(grid), dim3(kTileThreads), lds, stream, value,  \
                           lstart, src, grad_out, (float *)grad_value, (float *)grad_loc, (float *)grad_attn,        \
                           grad_proj, pl);                                                                           \
    } while (0)
                const bool b16 = sizeof(TV) == 2;
#define MSDA_LAUNCH_BINS(NI, NAME)                                                                                   \
    do {                                                                                                             \
        g_kernel = NAME;                                                                                             \
        hipLaunchKernelGGL((msda_bwd_d32_bins<NI, TV>), dim3(grid), dim3(kTileThreads), lds, stream, value, lstart,  \
                           src, grad_out, (float *)grad_value, (float *)grad_loc, (float *)grad_attn, grad_proj, pl, \
                           bp);                                                                                      \
    } while (0)
                if (variant == 12) {
                    if (bins_ni == 2) MSDA_LAUNCH_BINS(2, split ? (b16 ? "msda_bwd_d32_tile_bins<bf16,split>" : "msda_bwd_d32_tile_bins<split>")
                                                                : (b16 ? "msda_bwd_d32_tile_bins<bf16>" : "msda_bwd_d32_tile_bins"));
                    else MSDA_LAUNCH_BINS(3, split ? (b16 ? "msda_bwd_d32_tile_bins<3,bf16,split>" : "msda_bwd_d32_tile_bins<3,split>")
                                                   : (b16 ? "msda_bwd_d32_tile_bins<3,bf16>" : "msda_bwd_d32_tile_bins<3>"));
                } else if (variant == 11) {
                    if (split) MSDA_LAUNCH_LV(4, false, b16 ? "msda_bwd_d32_tile_lv<4,bf16,split>" : "msda_bwd_d32_tile_lv<4,split>");
                    else if (fused) MSDA_LAUNCH_LV(4, true, b16 ? "msda_bwd_d32_tile_lv<4,bf16,fused>" : "msda_bwd_d32_tile_lv<4,fused>");
                    else MSDA_LAUNCH_LV(4, false, b16 ? "msda_bwd_d32_tile_lv<4,bf16>" : "msda_bwd_d32_tile_lv<4>");
                } else {
                    if (split) MSDA_LAUNCH_LV(2, false, b16 ? "msda_bwd_d32_tile_lv<2,bf16,split>" : "msda_bwd_d32_tile_lv<2,split>");
                    else if (fused) MSDA_LAUNCH_LV(2, true, b16 ? "msda_bwd_d32_tile_lv<2,bf16,fused>" : "msda_bwd_d32_tile_lv<2,fused>");
                    else MSDA_LAUNCH_LV(2, false, b16 ? "msda_bwd_d32_tile_lv<2,bf16>" : "msda_bwd_d32_tile_lv<2>");
                }
#undef MSDA_LAUNCH_LV
#undef MSDA_LAUNCH_BINS
                rc = check_launch(g_kernel);
                if (rc || !fused) return rc;
                const int jgrid = clamp_grid((n_rows * 8 + 255) / 256, 16);
                if (split) {
                    if (rows16 && offsets_done) {
                        hipLaunchKernelGGL(msda_fused_finish16_rows_kernel, dim3(clamp_grid((n_rows + 255) / 256, 32)),
                                           dim3(256), 0, stream, src, (unsigned)n_rows, (unsigned)M, grad_proj);
                    } else if (L * P <= 16) {
                        hipLaunchKernelGGL(msda_fused_finish16_kernel, dim3(clamp_grid((n_rows * 16 + 255) / 256, 32)),
                                           dim3(256), 0, stream, shapes, src, n_rows, M, L, P, grad_proj, offsets_done);
                    } else {
                        hipLaunchKernelGGL(msda_fused_finish_kernel, dim3(jgrid), dim3(256), 0, stream, shapes, src,
                                           n_rows, M, L, P, grad_proj);
                    }
                    const char *name = g_kernel;
                    rc = check_launch("msda_fused_finish_kernel");
                    g_kernel = name;
                    return rc;
                }
                hipLaunchKernelGGL(msda_softmax_jacobian_kernel, dim3(jgrid), dim3(256), 0, stream, src, n_rows, M, L * P,
                                   grad_proj);
                const char *name = g_kernel;
                rc = check_launch("msda_softmax_jacobian_kernel");
                g_kernel = name;
                return rc;
            }
            variant = 8;
        }
        if (variant >= 2) {
            TilePlan pl;
            size_t lds = 0;
            const size_t rec_bytes = (size_t)32 * (2 * L * P + 1) * 16;
            if (make_tile_plan(pl, shapes_host, N, S, M, D, L, Lq, P, value_bytes, opt_bwd_tile_margin.load(), 0, 8,
                               rec_bytes, lds)) {
                const int grid = (pl.n_blocks + 7) & ~7;
                const PointSrc src = make_src(loc, attn, fa, M, L, P);
                pl.ablate = opt_bwd_ablate.load();
                pl.wide_log2 = opt_bwd_wide_log2.load();
#define MSDA_LAUNCH_TQ(PTS, FU, NAME)                                                                                \
    do {                                                                                                             \
        rc = allow_big_lds(msda_bwd_d32_tile_q2<PTS, TV, FU>, lds);                                                  \
        if (rc) return rc;                                                                                           \
        g_kernel = NAME;                                                                                             \
        hipLaunchKernelGGL((msda_bwd_d32_tile_q2<PTS, TV, FU>), dim3(grid), dim3(kTileThreads), lds, stream, value,  \
                           lstart, src, grad_out, (float *)grad_value, (float *)grad_loc, (float *)grad_attn,        \
                           grad_proj, pl);                                                                           \
    } while (0)
                const bool b16 = sizeof(TV) == 2;
                if (variant == 9) {
                    if (fused) MSDA_LAUNCH_TQ(4, true, b16 ? "msda_bwd_d32_tile_q2<4,bf16,fused>" : "msda_bwd_d32_tile_q2<4,fused>");
                    else MSDA_LAUNCH_TQ(4, false, b16 ? "msda_bwd_d32_tile_q2<4,bf16>" : "msda_bwd_d32_tile_q2<4>");
                } else {
                    if (fused) MSDA_LAUNCH_TQ(2, true, b16 ? "msda_bwd_d32_tile_q2<2,bf16,fused>" : "msda_bwd_d32_tile_q2<2,fused>");
                    else MSDA_LAUNCH_TQ(2, false, b16 ? "msda_bwd_d32_tile_q2<2,bf16>" : "msda_bwd_d32_tile_q2<2>");
                }
#undef MSDA_LAUNCH_TQ
                return check_launch(g_kernel);
            }
            variant = 1;
        }
    }
    const long n_rows = (long)N * Lq * M;
    if constexpr (kD32Type) {
        // few-query calls at D = 32 (the decoder's cross-attention): 32 lanes per row, grad_value atomics in whole
        // 128-byte rows, four points in flight (msda_bwd_rows.h)
        const int requested = opt_bwd_variant.load();      // 1 = the generic kernel, explicitly
        if (d32_ok(D, L, value_elems) && L * P <= kRowsMaxLP && opt_bwd_rows.load() != 0 && n_rows < (1L << 31) &&
            requested != 1) {
            const PointSrc src = make_src(loc, attn, fa, M, L, P);
            // a row is a chain of dependent round trips (stage -> loads -> atomics -> reductions): small calls get one
            // wavefront (two rows) per workgroup so that every CU holds several chains
            int threads = opt_bwd_rows_block.load();
            if (threads != 64 && threads != 128 && threads != 256) threads = n_rows <= 16384 ? 64 : 256;
            const int per = threads / 32;
            const int grid = clamp_grid((n_rows + per - 1) / per, 64);
            const unsigned gv_bytes = (unsigned)(value_elems * 4);
            const bool b16 = sizeof(TV) == 2;
            if (fused) {
                g_kernel = b16 ? "msda_bwd_d32_rows<bf16,fused>" : "msda_bwd_d32_rows<fused>";
                hipLaunchKernelGGL((msda_bwd_d32_rows<TV, true>), dim3(grid), dim3(threads), 0, stream, value, shapes, lstart,
                                   src, grad_out, N, S, M, L, Lq, P, (float *)grad_value, (float *)nullptr,
                                   (float *)nullptr, grad_proj, grad_ref_part, (unsigned)value_bytes, gv_bytes);
            } else {
                g_kernel = b16 ? "msda_bwd_d32_rows<bf16>" : "msda_bwd_d32_rows";
                hipLaunchKernelGGL((msda_bwd_d32_rows<TV, false>), dim3(grid), dim3(threads), 0, stream, value, shapes, lstart,
                                   src, grad_out, N, S, M, L, Lq, P, (float *)grad_value, (float *)grad_loc,
                                   (float *)grad_attn, (float *)nullptr, (float *)nullptr, (unsigned)value_bytes, gv_bytes);
            }
            return check_launch(g_kernel);
        }
    }
    int block = ((D + 63) / 64) * 64;
    if (block > 1024) block = 1024;
    const int grid = (int)(n_rows < 65536L * 16 ? n_rows : 65536L * 16);
    if constexpr (sizeof(TC) == 4) {
        if (fused) {
            const PointSrc src = make_src(loc, attn, fa, M, L, P);
            g_kernel = "msda_bwd_generic<fused>";
            hipLaunchKernelGGL((msda_bwd_generic<TV, TC, TG, true>), dim3(grid), dim3(block), 0, stream, value, shapes,
                               lstart, (const TC *)nullptr, (const TC *)nullptr, src, grad_out, N, S, M, D, L, Lq, P,
                               grad_value, (TC *)nullptr, (TC *)nullptr, grad_proj, grad_ref_part);
            return check_launch(g_kernel);
        }
    }
    g_kernel = "msda_bwd_generic";
    hipLaunchKernelGGL((msda_bwd_generic<TV, TC, TG, false>), dim3(grid), dim3(block), 0, stream, value, shapes, lstart,
                       loc, attn, PointSrc{}, grad_out, N, S, M, D, L, Lq, P, grad_value, grad_loc, grad_attn,
                       (float *)nullptr, (float *)nullptr);
    return check_launch(g_kernel);
}

}  // namespace

extern "C" {

int msda_abi_version(void) { return 4; }

void msda_set_call_site(uint64_t site) { g_site = site; }

int msda_selector_last(int *level, float *off_share, float *inner_share) {
    if (level) *level = g_sel_level;
    if (off_share) *off_share = g_sel_frac;
    if (inner_share) *inner_share = g_sel_inner;
    return MSDA_OK;
}

int msda_selector_next(int kind, int level, int off_permille, int inner_permille) {
    SelRule r;
    if (kind == 0) { r.up0 = opt_sel_fwd_up.load(); r.down1 = opt_sel_fwd_down.load(); r.up1 = r.down2 = 0; }
    else { r.up0 = opt_sel_up0.load(); r.up1 = opt_sel_up1.load(); r.down1 = opt_sel_down1.load(); r.down2 = opt_sel_down2.load(); }
    return sel_next_level(kind, level, (float)off_permille, (float)inner_permille, r);
}
const char *msda_last_error(void) { return g_err; }
const char *msda_last_kernel(void) { return g_kernel; }

int msda_forward_f32(const float *value, const int64_t *shapes_dev, const int64_t *lstart_dev, const float *loc,
                     const float *attn, int N, int S, int M, int D, int L, int Lq, int P, float *out,
                     const int64_t *shapes_host, void *stream) {
    return forward_impl<float, float>(value, shapes_dev, lstart_dev, loc, attn, FusedArgs{}, N, S, M, D, L, Lq, P, out,
                                      shapes_host, (hipStream_t)stream);
}

int msda_forward_f64(const double *value, const int64_t *shapes_dev, const int64_t *lstart_dev, const double *loc,
                     const double *attn, int N, int S, int M, int D, int L, int Lq, int P, double *out,
                     const int64_t *shapes_host, void *stream) {
    return forward_impl<double, double>(value, shapes_dev, lstart_dev, loc, attn, FusedArgs{}, N, S, M, D, L, Lq, P,
                                        out, shapes_host, (hipStream_t)stream);
}

int msda_forward_bf16(const uint16_t *value, const int64_t *shapes_dev, const int64_t *lstart_dev, const float *loc,
                      const float *attn, int N, int S, int M, int D, int L, int Lq, int P, uint16_t *out,
                      const int64_t *shapes_host, void *stream) {
    return forward_impl<bf16_t, float>((const bf16_t *)value, shapes_dev, lstart_dev, loc, attn, FusedArgs{}, N, S, M,
                                       D, L, Lq, P, (bf16_t *)out, shapes_host, (hipStream_t)stream);
}

int msda_backward_f32(const float *value, const int64_t *shapes_dev, const int64_t *lstart_dev, const float *loc,
                      const float *attn, const float *grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                      float *grad_value, float *grad_loc, float *grad_attn, int zero_grad_value,
                      const int64_t *shapes_host, void *stream) {
    return backward_impl<float, float, float>(value, shapes_dev, lstart_dev, loc, attn, FusedArgs{}, grad_out, N, S, M,
                                              D, L, Lq, P, grad_value, grad_loc, grad_attn, nullptr, nullptr,
                                              zero_grad_value, shapes_host, (hipStream_t)stream);
}

int msda_backward_f64(const double *value, const int64_t *shapes_dev, const int64_t *lstart_dev, const double *loc,
                      const double *attn, const double *grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                      double *grad_value, double *grad_loc, double *grad_attn, int zero_grad_value,
                      const int64_t *shapes_host, void *stream) {
    return backward_impl<double, double, double>(value, shapes_dev, lstart_dev, loc, attn, FusedArgs{}, grad_out, N, S,
                                                 M, D, L, Lq, P, grad_value, grad_loc, grad_attn, nullptr, nullptr,
                                                 zero_grad_value, shapes_host, (hipStream_t)stream);
}

int msda_backward_bf16(const uint16_t *value, const int64_t *shapes_dev, const int64_t *lstart_dev, const float *loc,
                       const float *attn, const uint16_t *grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                       float *grad_value, float *grad_loc, float *grad_attn, int zero_grad_value,
                       const int64_t *shapes_host, void *stream) {
    return backward_impl<bf16_t, float, float>((const bf16_t *)value, shapes_dev, lstart_dev, loc, attn, FusedArgs{},
                                               (const bf16_t *)grad_out, N, S, M, D, L, Lq, P, grad_value, grad_loc,
                                               grad_attn, nullptr, nullptr, zero_grad_value, shapes_host,
                                               (hipStream_t)stream);
}

// ---- fused prologue entry points ----
static FusedArgs fused_args(const float *proj, int proj_stride, const float *ref, int ref_dim, const uint8_t *mask) {
    FusedArgs fa;
    fa.proj = proj;
    fa.proj_stride = proj_stride;
    fa.ref = ref;
    fa.ref_dim = ref_dim;
    fa.mask = mask;
    return fa;
}

int msda_fused_forward_f32(const float *value, const int64_t *shapes_dev, const int64_t *lstart_dev, const float *proj,
                           int proj_stride, const float *ref, int ref_dim, const uint8_t *pad_mask, int N, int S, int M,
                           int D, int L, int Lq, int P, float *out, const int64_t *shapes_host, void *stream) {
    if (!proj) return fail(MSDA_EINVAL, "null pointer argument");
    return forward_impl<float, float>(value, shapes_dev, lstart_dev, nullptr, nullptr,
                                      fused_args(proj, proj_stride, ref, ref_dim, pad_mask), N, S, M, D, L, Lq, P, out,
                                      shapes_host, (hipStream_t)stream);
}

int msda_fused_forward_bf16(const uint16_t *value, const int64_t *shapes_dev, const int64_t *lstart_dev,
                            const float *proj, int proj_stride, const float *ref, int ref_dim, const uint8_t *pad_mask,
                            int N, int S, int M, int D, int L, int Lq, int P, uint16_t *out,
                            const int64_t *shapes_host, void *stream) {
    if (!proj) return fail(MSDA_EINVAL, "null pointer argument");
    return forward_impl<bf16_t, float>((const bf16_t *)value, shapes_dev, lstart_dev, nullptr, nullptr,
                                       fused_args(proj, proj_stride, ref, ref_dim, pad_mask), N, S, M, D, L, Lq, P,
                                       (bf16_t *)out, shapes_host, (hipStream_t)stream);
}

int msda_fused_backward_f32(const float *value, const int64_t *shapes_dev, const int64_t *lstart_dev,
                            const float *proj, int proj_stride, const float *ref, int ref_dim, const uint8_t *pad_mask,
                            const float *grad_out, int N, int S, int M, int D, int L, int Lq, int P, float *grad_value,
                            float *grad_proj, float *grad_ref_part, int zero_grad_value, const int64_t *shapes_host,
                            void *stream) {
    if (!proj) return fail(MSDA_EINVAL, "null pointer argument");
    return backward_impl<float, float, float>(value, shapes_dev, lstart_dev, nullptr, nullptr,
                                              fused_args(proj, proj_stride, ref, ref_dim, pad_mask), grad_out, N, S, M,
                                              D, L, Lq, P, grad_value, nullptr, nullptr, grad_proj, grad_ref_part,
                                              zero_grad_value, shapes_host, (hipStream_t)stream);
}

int msda_fused_backward_bf16(const uint16_t *value, const int64_t *shapes_dev, const int64_t *lstart_dev,
                             const float *proj, int proj_stride, const float *ref, int ref_dim, const uint8_t *pad_mask,
                             const uint16_t *grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                             float *grad_value, float *grad_proj, float *grad_ref_part, int zero_grad_value,
                             const int64_t *shapes_host, void *stream) {
    if (!proj) return fail(MSDA_EINVAL, "null pointer argument");
    return backward_impl<bf16_t, float, float>((const bf16_t *)value, shapes_dev, lstart_dev, nullptr, nullptr,
                                               fused_args(proj, proj_stride, ref, ref_dim, pad_mask),
                                               (const bf16_t *)grad_out, N, S, M, D, L, Lq, P, grad_value, nullptr,
                                               nullptr, grad_proj, grad_ref_part, zero_grad_value, shapes_host,
                                               (hipStream_t)stream);
}

size_t msda_fused_workspace_bytes(int N, int Lq, int M, int L, int P) {
    if (N < 0 || Lq < 0 || M <= 0 || L <= 0 || P <= 0) return 0;
    return (size_t)N * Lq * M * L * P * 3 * sizeof(float);
}

int msda_fused_backward_ws_f32(const float *value, const int64_t *shapes_dev, const int64_t *lstart_dev,
                               const float *proj, int proj_stride, const float *ref, int ref_dim,
                               const uint8_t *pad_mask, const float *grad_out, int N, int S, int M, int D, int L, int Lq,
                               int P, float *grad_value, float *grad_proj, float *grad_ref_part, int zero_grad_value,
                               const int64_t *shapes_host, void *workspace, size_t workspace_bytes, void *stream) {
    if (!proj) return fail(MSDA_EINVAL, "null pointer argument");
    return backward_impl<float, float, float>(value, shapes_dev, lstart_dev, nullptr, nullptr,
                                              fused_args(proj, proj_stride, ref, ref_dim, pad_mask), grad_out, N, S, M,
                                              D, L, Lq, P, grad_value, nullptr, nullptr, grad_proj, grad_ref_part,
                                              zero_grad_value, shapes_host, (hipStream_t)stream, (float *)workspace,
                                              workspace_bytes);
}

int msda_fused_backward_ws_bf16(const uint16_t *value, const int64_t *shapes_dev, const int64_t *lstart_dev,
                                const float *proj, int proj_stride, const float *ref, int ref_dim,
                                const uint8_t *pad_mask, const uint16_t *grad_out, int N, int S, int M, int D, int L,
                                int Lq, int P, float *grad_value, float *grad_proj, float *grad_ref_part,
                                int zero_grad_value, const int64_t *shapes_host, void *workspace,
                                size_t workspace_bytes, void *stream) {
    if (!proj) return fail(MSDA_EINVAL, "null pointer argument");
    return backward_impl<bf16_t, float, float>((const bf16_t *)value, shapes_dev, lstart_dev, nullptr, nullptr,
                                               fused_args(proj, proj_stride, ref, ref_dim, pad_mask),
                                               (const bf16_t *)grad_out, N, S, M, D, L, Lq, P, grad_value, nullptr,
                                               nullptr, grad_proj, grad_ref_part, zero_grad_value, shapes_host,
                                               (hipStream_t)stream, (float *)workspace, workspace_bytes);
}

int msda_fused_points_f32(const int64_t *shapes_dev, const float *proj, int proj_stride, const float *ref, int ref_dim,
                          int N, int M, int L, int Lq, int P, float *loc_out, float *attn_out, void *stream) {
    if (!shapes_dev || !proj || !ref || !loc_out || !attn_out) return fail(MSDA_EINVAL, "null pointer argument");
    if (N < 0 || Lq < 0 || M <= 0 || L <= 0 || P <= 0) return fail(MSDA_EINVAL, "bad dimension");
    const FusedArgs fa = fused_args(proj, proj_stride, ref, ref_dim, nullptr);
    const int rc = check_fused(fa, M, L, P);
    if (rc) return rc;
    const long n_rows = (long)N * Lq * M;
    if (n_rows == 0) { g_err[0] = 0; return MSDA_OK; }
    const PointSrc src = make_src(nullptr, nullptr, fa, M, L, P);
    if (L * P <= 16) {
        hipLaunchKernelGGL(msda_fused_points16_kernel, dim3(clamp_grid((n_rows * 16 + 255) / 256, 32)), dim3(256), 0,
                           (hipStream_t)stream, shapes_dev, src, n_rows, M, L, P, loc_out, attn_out);
        return check_launch("msda_fused_points16_kernel");
    }
    const int grid = clamp_grid((n_rows * 8 + 255) / 256, 16);
    hipLaunchKernelGGL(msda_fused_points_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, shapes_dev, src, n_rows,
                       M, L, P, loc_out, attn_out);
    return check_launch("msda_fused_points_kernel");
}

int msda_sample_indices_f32(const int64_t *shapes_dev, const float *loc, int N, int M, int L, int Lq, int P,
                            int32_t *h_low, int32_t *w_low, uint8_t *gate, void *stream) {
    if (!shapes_dev || !loc || !h_low || !w_low || !gate) return fail(MSDA_EINVAL, "null pointer argument");
    if (N < 0 || Lq < 0 || M <= 0 || L <= 0 || P <= 0) return fail(MSDA_EINVAL, "bad dimension");
    const long n_points = (long)N * Lq * M * L * P;
    if (n_points == 0) { g_err[0] = 0; return MSDA_OK; }
    const int grid = clamp_grid((n_points + 255) / 256, 16);
    hipLaunchKernelGGL(msda_indices_f32_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, shapes_dev, loc,
                       n_points, L, P, h_low, w_low, gate);
    return check_launch("msda_indices_f32_kernel");
}

static std::atomic<int> *find_opt(const char *key) {
    if (!key) return nullptr;
    if (!strcmp(key, "fwd_variant")) return &opt_fwd_variant;
    if (!strcmp(key, "bwd_variant")) return &opt_bwd_variant;
    if (!strcmp(key, "fwd_block")) return &opt_fwd_block;
    if (!strcmp(key, "bwd_block")) return &opt_bwd_block;
    if (!strcmp(key, "fwd_grid_mult")) return &opt_fwd_grid_mult;
    if (!strcmp(key, "bwd_grid_mult")) return &opt_bwd_grid_mult;
    if (!strcmp(key, "fwd_tile_margin")) return &opt_fwd_tile_margin;
    if (!strcmp(key, "bwd_tile_margin")) return &opt_bwd_tile_margin;
    if (!strcmp(key, "fwd_tile_l0")) return &opt_fwd_tile_l0;
    if (!strcmp(key, "bwd_ablate")) return &opt_bwd_ablate;
    if (!strcmp(key, "bwd_wide_log2")) return &opt_bwd_wide_log2;
    if (!strcmp(key, "bwd_split")) return &opt_bwd_split;
    if (!strcmp(key, "bwd_bins_margin")) return &opt_bwd_bins_margin;
    if (!strcmp(key, "bwd_bins_margin_hi")) return &opt_bwd_bins_margin_hi;
    if (!strcmp(key, "auto_select")) return &opt_auto_select;
    if (!strcmp(key, "sel_level")) return &opt_sel_level;
    if (!strcmp(key, "sel_up0")) return &opt_sel_up0;
    if (!strcmp(key, "sel_up1")) return &opt_sel_up1;
    if (!strcmp(key, "sel_down1")) return &opt_sel_down1;
    if (!strcmp(key, "sel_down2")) return &opt_sel_down2;
    if (!strcmp(key, "sel_fwd_up")) return &opt_sel_fwd_up;
    if (!strcmp(key, "sel_fwd_down")) return &opt_sel_fwd_down;
    if (!strcmp(key, "bwd_bins_strip")) return &opt_bwd_bins_strip;
    if (!strcmp(key, "bwd_rows")) return &opt_bwd_rows;
    if (!strcmp(key, "bwd_rows_block")) return &opt_bwd_rows_block;
    if (!strcmp(key, "fwd_win_rlog")) return &opt_fwd_win_rlog;
    if (!strcmp(key, "fwd_win_rlogx")) return &opt_fwd_win_rlogx;
    if (!strcmp(key, "fwd_win_auto")) return &opt_fwd_win_auto;
    if (!strcmp(key, "fwd_head_major")) return &opt_fwd_head_major;
    if (!strcmp(key, "fwd_win_block")) return &opt_fwd_win_block;
    if (!strcmp(key, "fwd_win_l0")) return &opt_fwd_win_l0;
    if (!strcmp(key, "fwd_win_margins")) return &opt_fwd_win_margins;
    if (!strcmp(key, "fwd_win_dma")) return &opt_fwd_win_dma;
    if (!strcmp(key, "bwd_side_rows")) return &opt_bwd_side_rows;
    if (!strcmp(key, "fwd_win_trace_lo")) return &opt_fwd_win_trace_lo;
    if (!strcmp(key, "fwd_win_trace_hi")) return &opt_fwd_win_trace_hi;
    if (!strcmp(key, "fwd_win_ablate")) return &opt_fwd_win_ablate;
    if (!strcmp(key, "fwd_win_wps")) return &opt_fwd_win_wps;
    if (!strcmp(key, "fwd_win_early")) return &opt_fwd_win_early;
    return nullptr;
}

int msda_set_option(const char *key, int value) {
    std::atomic<int> *o = find_opt(key);
    if (!o || (value < 0 && !(o == &opt_sel_level && value == -1)))
        return fail(MSDA_EINVAL, "unknown option or negative value");
    o->store(value);
    return MSDA_OK;
}

int msda_get_option(const char *key, int *value) {
    std::atomic<int> *o = find_opt(key);
    if (!o || !value) return fail(MSDA_EINVAL, "unknown option");
    *value = o->load();
    return MSDA_OK;
}

}  // extern "C"
