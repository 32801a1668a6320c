// msda_hip.hip -- multi-scale deformable attention for gfx950 (MI355X, CDNA4).
//
// Hand-written HIP; wave64, LDS-staged sampling records, buffer (SRSRC) gathers with
// hardware zero padding, DPP reductions, hardware f32 atomics.  No CUDA-compat layer.
//
// Semantics replaced (reference repository paths):
//   forward   models/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299 (+ bilinear :33-84)
//   backward  models/ops/src/cuda/ms_deform_im2col_cuda.cuh:301-403 (+ bilinear :87-159)
//   host side models/ops/src/cuda/ms_deform_attn_cuda.cu:20-153
// C ABI: include/msda_hip.h.  Design notes, byte counts and rooflines: DESIGN.md.
//
// Kernel families
//   *_generic   any D/L/P, f32 / f64 / bf16 storage: one thread per output scalar
//               (forward) or one block per (n,q,m) row (backward).  Correctness path for
//               shapes the specialised kernels do not cover (reference gradcheck sizes
//               D in {30,64,71,1025,...}).
//   *_d32       MeMOTR geometry (D = 32 channels/head, fp32): 8 lanes x float4 own one
//               (n,q,m) row; a wavefront owns 8 rows.  Each lane prepares the sampling
//               record (4 corner byte offsets + 4 fused weights) of 1/8 of the row's
//               L*P points exactly once, parks it in LDS, and the 8 lanes of the row
//               then stream the records back as broadcast ds_read_b128.  Corner reads
//               are 128-byte buffer_load_dwordx4 rows; invalid corners carry an
//               out-of-range offset so the buffer unit returns zeros (= the reference's
//               per-corner zero padding, no divergent branches).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>

#include "../../include/msda_hip.h"

namespace {

constexpr int kWave = 64;
constexpr int kMaxLevels = 16;          // level table kept in LDS by the specialised kernels
constexpr unsigned kOobOffset = 0x80000000u;  // >= any legal byte offset (tensors < 2 GiB)
constexpr int kNumCU = 256;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ----------------------------------------------------------------------------------------
// storage <-> compute conversions
// ----------------------------------------------------------------------------------------
struct bf16_t {
    uint16_t bits;
};

__device__ __forceinline__ float to_compute(float x) { return x; }
__device__ __forceinline__ double to_compute(double x) { return x; }
__device__ __forceinline__ float to_compute(bf16_t x) { return __uint_as_float(((unsigned)x.bits) << 16); }

template <typename TS, typename TC>
__device__ __forceinline__ TS to_storage(TC x);
template <>
__device__ __forceinline__ float to_storage<float, float>(float x) { return x; }
template <>
__device__ __forceinline__ double to_storage<double, double>(double x) { return x; }
template <>
__device__ __forceinline__ bf16_t to_storage<bf16_t, float>(float x) {
    unsigned u = __float_as_uint(x);
    bf16_t r;
    if ((u & 0x7fffffffu) > 0x7f800000u) {  // NaN: keep it quiet
        r.bits = (uint16_t)((u >> 16) | 0x0040u);
    } else {
        u += 0x7fffu + ((u >> 16) & 1u);  // round to nearest even
        r.bits = (uint16_t)(u >> 16);
    }
    return r;
}

// ----------------------------------------------------------------------------------------
// The sampling arithmetic shared by every kernel.  Rounding points are the reference's:
// the product loc*size is rounded to T first, then 0.5 is subtracted (no FMA contraction),
// so floor() and the gate see exactly the reference's h_im / w_im (.cuh:285-288).
// ----------------------------------------------------------------------------------------
template <typename T>
struct Sample {
    bool gate;
    int h_low, w_low;
    T lh, lw;
};

template <typename T>
__device__ __forceinline__ Sample<T> sample_setup(T loc_w, T loc_h, int H, int W) {
#pragma clang fp contract(off)
    Sample<T> s;
    const T hf = (T)H, wf = (T)W;
    const T ph = loc_h * hf;
    const T pw = loc_w * wf;
    const T h_im = ph - (T)0.5;
    const T w_im = pw - (T)0.5;
    s.gate = (h_im > (T)-1) && (w_im > (T)-1) && (h_im < hf) && (w_im < wf);
    const T fh = floor(h_im), fw = floor(w_im);
    s.h_low = (int)fh;
    s.w_low = (int)fw;
    s.lh = h_im - fh;
    s.lw = w_im - fw;
    return s;
}

template <typename T>
__device__ __forceinline__ void atomic_add_hw(T *p, T v) {
    unsafeAtomicAdd(p, v);  // global_atomic_add_f32 / _f64, no CAS loop
}

// ----------------------------------------------------------------------------------------
// generic forward: one thread per output scalar (n,q,m,c); consecutive threads walk c.
// ----------------------------------------------------------------------------------------
template <typename TV, typename TC>
__global__ __launch_bounds__(256) void msda_fwd_generic(
    const TV *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lstart,
    const TC *__restrict__ loc, const TC *__restrict__ attn, int N, int S, int M, int D, int L, int Lq, int P,
    TV *__restrict__ out) {
    const long total = (long)N * Lq * M * D;
    const long row = (long)M * D;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % D);
        const long pm = idx / D;
        const int m = (int)(pm % M);
        const long b = pm / M / Lq;
        const TC *lp = loc + pm * L * P * 2;
        const TC *ap = attn + pm * L * P;
        TC acc = (TC)0;
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            const TV *v = value + (b * S + lstart[l]) * row + (long)m * D + c;
            for (int p = 0; p < P; ++p) {
                const Sample<TC> s = sample_setup<TC>(lp[0], lp[1], H, W);
                const TC a = ap[0];
                lp += 2;
                ap += 1;
                if (!s.gate) continue;
                const TC hh = (TC)1 - s.lh, hw = (TC)1 - s.lw;
                const int h0 = s.h_low, w0 = s.w_low, h1 = h0 + 1, w1 = w0 + 1;
                TC v1 = 0, v2 = 0, v3 = 0, v4 = 0;
                if (h0 >= 0 && w0 >= 0) v1 = to_compute(v[((long)h0 * W + w0) * row]);
                if (h0 >= 0 && w1 <= W - 1) v2 = to_compute(v[((long)h0 * W + w1) * row]);
                if (h1 <= H - 1 && w0 >= 0) v3 = to_compute(v[((long)h1 * W + w0) * row]);
                if (h1 <= H - 1 && w1 <= W - 1) v4 = to_compute(v[((long)h1 * W + w1) * row]);
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
// ----------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, kWave);
    return x;
}

template <typename TV, typename TC, typename TG>
__global__ __launch_bounds__(1024) void msda_bwd_generic(
    const TV *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lstart,
    const TC *__restrict__ loc, const TC *__restrict__ attn, const TV *__restrict__ grad_out, int N, int S, int M,
    int D, int L, int Lq, int P, TG *__restrict__ grad_value, TC *__restrict__ grad_loc, TC *__restrict__ grad_attn) {
    __shared__ TC red[3 * 16];
    const long n_rows = (long)N * Lq * M;
    const long row = (long)M * D;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    for (long pm = blockIdx.x; pm < n_rows; pm += gridDim.x) {
        const int m = (int)(pm % M);
        const long b = pm / M / Lq;
        const TV *g = grad_out + pm * D;
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            const long base = (b * S + lstart[l]) * row + (long)m * D;
            for (int p = 0; p < P; ++p) {
                const long t = (pm * L + l) * P + p;
                const Sample<TC> s = sample_setup<TC>(loc[2 * t], loc[2 * t + 1], H, W);
                TC acc_w = 0, acc_h = 0, acc_a = 0;
                if (s.gate) {  // block-uniform
                    const TC a = attn[t];
                    const TC hh = (TC)1 - s.lh, hw = (TC)1 - s.lw;
                    const TC w1 = hh * hw, w2 = hh * s.lw, w3 = s.lh * hw, w4 = s.lh * s.lw;
                    const int h0 = s.h_low, w0 = s.w_low, h1 = h0 + 1, w1i = w0 + 1;
                    const bool ok1 = (h0 >= 0 && w0 >= 0), ok2 = (h0 >= 0 && w1i <= W - 1);
                    const bool ok3 = (h1 <= H - 1 && w0 >= 0), ok4 = (h1 <= H - 1 && w1i <= W - 1);
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
                    grad_loc[2 * t] = acc_w;
                    grad_loc[2 * t + 1] = acc_h;
                    grad_attn[t] = acc_a;
                }
                if (n_waves > 1) __syncthreads();
            }
        }
    }
}

// ----------------------------------------------------------------------------------------
// index probe (parity hook): same sample_setup as the kernels.
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

// ----------------------------------------------------------------------------------------
// D = 32, fp32 specialised kernels.
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes) {
    // raw buffer (stride 0), DATA_FORMAT = 32-bit; reads past `bytes` return 0, atomics/stores past it are dropped
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ f32x4 buf_load_f4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}

// Sampling record of one (row, point): 4 corner byte offsets (relative to the tensor base,
// kOobOffset when the corner is outside the level or the point is gated off) + 4 floats.
struct PointRec {
    u32x4 off;
    f32x4 w;
};

// XCD-aware task walk: hardware places block b on XCD b % 8 (observed; speed only).  Give each
// XCD one contiguous eighth of the (raster-ordered) rows so the level slabs it touches stay in
// its private 4 MiB L2.
struct TaskWalk {
    long begin, end, step;
};
__device__ __forceinline__ TaskWalk xcd_walk(long n_tasks, int waves_per_block) {
    TaskWalk w;
    const int xcd = blockIdx.x & 7;
    const int blk_in_xcd = blockIdx.x >> 3;
    const int blks_per_xcd = (gridDim.x + 7 - xcd) >> 3;  // blocks with this residue
    const long per = (n_tasks + 7) >> 3;
    const long lo = per * xcd;
    long hi = lo + per;
    if (hi > n_tasks) hi = n_tasks;
    w.begin = lo + (long)blk_in_xcd * waves_per_block + (threadIdx.x >> 6);
    w.end = hi;
    w.step = (long)blks_per_xcd * waves_per_block;
    return w;
}

// One chunk of PTS points of one row: all 4*PTS corner loads are issued before the first
// FMA so a wave keeps 4*PTS 128-byte requests in flight.
template <int PTS>
__device__ __forceinline__ void fwd_gather_chunk(const u32x4 *rec, int t0, __amdgpu_buffer_rsrc_t vr,
                                                 unsigned lane_off, f32x4 &acc) {
    u32x4 o[PTS];
    f32x4 w[PTS];
    f32x4 v[PTS][4];
#pragma unroll
    for (int i = 0; i < PTS; ++i) {
        o[i] = rec[2 * (t0 + i)];
        w[i] = __builtin_bit_cast(f32x4, rec[2 * (t0 + i) + 1]);
    }
#pragma unroll
    for (int i = 0; i < PTS; ++i) {
        v[i][0] = buf_load_f4(vr, o[i].x + lane_off);
        v[i][1] = buf_load_f4(vr, o[i].y + lane_off);
        v[i][2] = buf_load_f4(vr, o[i].z + lane_off);
        v[i][3] = buf_load_f4(vr, o[i].w + lane_off);
    }
#pragma unroll
    for (int i = 0; i < PTS; ++i) {
        acc += w[i].x * v[i][0];
        acc += w[i].y * v[i][1];
        acc += w[i].z * v[i][2];
        acc += w[i].w * v[i][3];
    }
}

// Prepare the LP/8 sampling records this lane owns for its row and park them in LDS.
// BWD = false: weights are the four corner weights pre-multiplied by the attention weight.
// BWD = true : record carries (lh, lw, attn, 0) -- the backward needs the factors apart.
template <bool BWD>
__device__ __forceinline__ void stage_records(u32x4 *rec, const float *__restrict__ loc,
                                              const float *__restrict__ attn, long pmc, bool row_ok, int sub, int LP,
                                              int P, int M, unsigned row_base, const int *s_H, const int *s_W,
                                              const int *s_start) {
    constexpr int D = 32;
    for (int t = sub; t < LP; t += 8) {
        const int l = t / P;
        const f32x2 xy = *reinterpret_cast<const f32x2 *>(loc + (pmc * LP + t) * 2);
        const float a = attn[pmc * LP + t];
        const int H = s_H[l], W = s_W[l];
        const Sample<float> s = sample_setup<float>(xy.x, xy.y, H, W);
        const float hh = 1.f - s.lh, hw = 1.f - s.lw;
        const int h0 = s.h_low, w0 = s.w_low, h1 = h0 + 1, w1 = w0 + 1;
        const bool live = s.gate && row_ok;
        const bool okh0 = live && h0 >= 0, okh1 = live && h1 <= H - 1;
        const bool okw0 = w0 >= 0, okw1 = w1 <= W - 1;
        const unsigned pix_stride = (unsigned)M * (D * 4u);
        const unsigned o00 = row_base + (unsigned)(s_start[l] + h0 * W + w0) * pix_stride;
        u32x4 off;
        off.x = (okh0 && okw0) ? o00 : kOobOffset;
        off.y = (okh0 && okw1) ? o00 + pix_stride : kOobOffset;
        off.z = (okh1 && okw0) ? o00 + (unsigned)W * pix_stride : kOobOffset;
        off.w = (okh1 && okw1) ? o00 + (unsigned)W * pix_stride + pix_stride : kOobOffset;
        f32x4 w;
        if (BWD) {
            w.x = s.lh;
            w.y = s.lw;
            w.z = row_ok ? a : 0.f;
            w.w = 0.f;
        } else {
            w.x = (hh * hw) * a;
            w.y = (hh * s.lw) * a;
            w.z = (s.lh * hw) * a;
            w.w = (s.lh * s.lw) * a;
        }
        rec[2 * t] = off;
        rec[2 * t + 1] = __builtin_bit_cast(u32x4, w);
    }
}

// forward, variants 2/3: direct gather (every corner row is read through the vector L1).
// PTS = points whose corner loads are kept in flight together.
template <int PTS>
__global__ __launch_bounds__(256) void msda_fwd_d32_gather(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lstart,
    const float *__restrict__ loc, const float *__restrict__ attn, int N, int S, int M, int L, int Lq, int P,
    float *__restrict__ out, unsigned value_bytes) {
    constexpr int D = 32;
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
    const int grp = lane >> 3, sub = lane & 7;
    const int rec_stride = 2 * LP + 1;  // in 16-byte units; +1 staggers the 8 rows over LDS banks
    u32x4 *rec = reinterpret_cast<u32x4 *>(s_dyn) + (size_t)(wave * 8 + grp) * rec_stride;
    const long n_rows = (long)N * Lq * M;
    const long n_tasks = (n_rows + 7) >> 3;
    const __amdgpu_buffer_rsrc_t vr = make_rsrc(value, value_bytes);
    const unsigned lane_off = (unsigned)sub * 16u;
    const TaskWalk tw = xcd_walk(n_tasks, wpb);
    for (long task = tw.begin; task < tw.end; task += tw.step) {
        const long pm = task * 8 + grp;
        const bool row_ok = pm < n_rows;
        const long pmc = row_ok ? pm : n_rows - 1;
        const int m = (int)(pmc % M);
        const int b = (int)(pmc / M / Lq);
        const unsigned row_base = ((unsigned)b * (unsigned)S * (unsigned)M + (unsigned)m) * (D * 4u);
        stage_records<false>(rec, loc, attn, pmc, row_ok, sub, LP, P, M, row_base, s_H, s_W, s_start);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int t = 0;
        for (; t + PTS <= LP; t += PTS) fwd_gather_chunk<PTS>(rec, t, vr, lane_off, acc);
        for (; t < LP; ++t) fwd_gather_chunk<1>(rec, t, vr, lane_off, acc);
        if (row_ok) *reinterpret_cast<f32x4 *>(out + pm * D + sub * 4) = acc;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// DPP butterfly over the 8 lanes that own one row.
__device__ __forceinline__ float sum8(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));  // row_half_mirror
    return x;
}

// One chunk of PTS points of one row, backward.  ATOMICS=false is an ablation build (no
// grad_value traffic) used only by tools/msda_bench to price the atomics; never auto-selected.
template <int PTS, bool ATOMICS>
__device__ __forceinline__ void bwd_chunk(const u32x4 *rec, float *res, int t0, int LP, int P, int sub,
                                          __amdgpu_buffer_rsrc_t vr, __amdgpu_buffer_rsrc_t gr, unsigned lane_off,
                                          const f32x4 g, const int *s_H, const int *s_W) {
    u32x4 o[PTS];
    f32x4 rw[PTS];
    f32x4 v[PTS][4];
#pragma unroll
    for (int i = 0; i < PTS; ++i) {
        o[i] = rec[2 * (t0 + i)];
        rw[i] = __builtin_bit_cast(f32x4, rec[2 * (t0 + i) + 1]);
    }
#pragma unroll
    for (int i = 0; i < PTS; ++i) {
        v[i][0] = buf_load_f4(vr, o[i].x + lane_off);
        v[i][1] = buf_load_f4(vr, o[i].y + lane_off);
        v[i][2] = buf_load_f4(vr, o[i].z + lane_off);
        v[i][3] = buf_load_f4(vr, o[i].w + lane_off);
    }
#pragma unroll
    for (int i = 0; i < PTS; ++i) {
        const int t = t0 + i;
        const float lh = rw[i].x, lw = rw[i].y, a = rw[i].z;
        const float hh = 1.f - lh, hw = 1.f - lw;
        const f32x4 tga = g * a;
        const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
        if (ATOMICS) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(w1 * tga[c], gr, (int)(o[i].x + lane_off + 4u * c), 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(w2 * tga[c], gr, (int)(o[i].y + lane_off + 4u * c), 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(w3 * tga[c], gr, (int)(o[i].z + lane_off + 4u * c), 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(w4 * tga[c], gr, (int)(o[i].w + lane_off + 4u * c), 0, 0);
            }
        }
        const f32x4 val = w1 * v[i][0] + w2 * v[i][1] + w3 * v[i][2] + w4 * v[i][3];
        const f32x4 gw = hh * (v[i][1] - v[i][0]) + lh * (v[i][3] - v[i][2]);
        const f32x4 gh = hw * (v[i][2] - v[i][0]) + lw * (v[i][3] - v[i][1]);
        float pa = g.x * val.x + g.y * val.y + g.z * val.z + g.w * val.w;
        float pw = gw.x * tga.x + gw.y * tga.y + gw.z * tga.z + gw.w * tga.w;
        float ph = gh.x * tga.x + gh.y * tga.y + gh.z * tga.z + gh.w * tga.w;
        pa = sum8(pa);
        pw = sum8(pw);
        ph = sum8(ph);
        if (sub == (t & 7)) {
            const int l = t / P;
            res[2 * t] = pw * (float)s_W[l];
            res[2 * t + 1] = ph * (float)s_H[l];
            res[2 * LP + t] = pa;
        }
    }
}

// backward, variants 2/3: same row ownership as the forward gather; grad_value via buffer
// atomics (dropped by hardware for out-of-range corners), channel sums via DPP, results
// parked in LDS and written back as coalesced rows.
template <int PTS, bool ATOMICS>
__global__ __launch_bounds__(256) void msda_bwd_d32_gather(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lstart,
    const float *__restrict__ loc, const float *__restrict__ attn, const float *__restrict__ grad_out, int N, int S,
    int M, int L, int Lq, int P, float *__restrict__ grad_value, float *__restrict__ grad_loc,
    float *__restrict__ grad_attn, unsigned value_bytes) {
    constexpr int D = 32;
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
    const int grp = lane >> 3, sub = lane & 7;
    const int rec_stride = 2 * LP + 1;
    u32x4 *rec = reinterpret_cast<u32x4 *>(s_dyn) + (size_t)(wave * 8 + grp) * rec_stride;
    // per-row result slots (grad_loc.x, grad_loc.y, grad_attn per point), 3*LP floats per row, padded
    float *res = reinterpret_cast<float *>(reinterpret_cast<u32x4 *>(s_dyn) + (size_t)wpb * 8 * rec_stride) +
                 (size_t)(wave * 8 + grp) * (3 * LP + 1);
    const long n_rows = (long)N * Lq * M;
    const long n_tasks = (n_rows + 7) >> 3;
    const __amdgpu_buffer_rsrc_t vr = make_rsrc(value, value_bytes);
    const __amdgpu_buffer_rsrc_t gr = make_rsrc(grad_value, value_bytes);
    const unsigned lane_off = (unsigned)sub * 16u;
    const TaskWalk tw = xcd_walk(n_tasks, wpb);
    for (long task = tw.begin; task < tw.end; task += tw.step) {
        const long pm = task * 8 + grp;
        const bool row_ok = pm < n_rows;
        const long pmc = row_ok ? pm : n_rows - 1;
        const int m = (int)(pmc % M);
        const int b = (int)(pmc / M / Lq);
        const unsigned row_base = ((unsigned)b * (unsigned)S * (unsigned)M + (unsigned)m) * (D * 4u);
        stage_records<true>(rec, loc, attn, pmc, row_ok, sub, LP, P, M, row_base, s_H, s_W, s_start);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const f32x4 g = *reinterpret_cast<const f32x4 *>(grad_out + pmc * D + sub * 4);
        int t = 0;
        for (; t + PTS <= LP; t += PTS) bwd_chunk<PTS, ATOMICS>(rec, res, t, LP, P, sub, vr, gr, lane_off, g, s_H, s_W);
        for (; t < LP; ++t) bwd_chunk<1, ATOMICS>(rec, res, t, LP, P, sub, vr, gr, lane_off, g, s_H, s_W);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (row_ok) {
            for (int i = sub; i < 2 * LP; i += 8) grad_loc[pm * LP * 2 + i] = res[i];
            for (int i = sub; i < LP; i += 8) grad_attn[pm * LP + i] = res[2 * LP + i];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ----------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------
thread_local char g_err[256] = {0};
thread_local const char *g_kernel = "";

std::atomic<int> opt_fwd_variant{0}, opt_bwd_variant{0};
std::atomic<int> opt_fwd_block{256}, opt_bwd_block{256};
std::atomic<int> opt_fwd_grid_mult{8}, opt_bwd_grid_mult{8};

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

bool d32_ok(int D, int L, long value_bytes) { return D == 32 && L <= kMaxLevels && value_bytes < 0x7fffff00L; }

template <typename TV, typename TC>
int forward_impl(const TV *value, const int64_t *shapes, const int64_t *lstart, const TC *loc, const TC *attn, int N,
                 int S, int M, int D, int L, int Lq, int P, TV *out, hipStream_t stream, bool allow_d32) {
    int rc = check_dims(value, shapes, lstart, loc, attn, out, N, S, M, D, L, Lq, P);
    if (rc) return rc;
    if ((long)N * Lq == 0) { g_err[0] = 0; return MSDA_OK; }
    int variant = opt_fwd_variant.load();
    const long value_bytes = (long)N * S * M * D * (long)sizeof(TV);
    const bool can32 = allow_d32 && d32_ok(D, L, value_bytes);
    if (variant == 0) variant = can32 ? 2 : 1;
    if (variant >= 2 && !can32) variant = 1;
    if (variant == 1) {
        const long total = (long)N * Lq * M * D;
        const int grid = clamp_grid((total + 255) / 256, 32);
        g_kernel = "msda_fwd_generic";
        hipLaunchKernelGGL((msda_fwd_generic<TV, TC>), dim3(grid), dim3(256), 0, stream, value, shapes, lstart, loc,
                           attn, N, S, M, D, L, Lq, P, out);
        return check_launch("msda_fwd_generic");
    }
    if constexpr (sizeof(TV) == 4 && sizeof(TC) == 4) {
        int block = opt_fwd_block.load();
        if (block < 64 || block > 1024 || (block & 63)) block = 256;
        const int wpb = block / 64;
        const long n_tasks = ((long)N * Lq * M + 7) / 8;
        // small problems: one wave per block so every task gets its own CU slot
        int use_block = block;
        if (n_tasks < (long)kNumCU * wpb) use_block = 64;
        const int uwpb = use_block / 64;
        int grid = clamp_grid((n_tasks + uwpb - 1) / uwpb, opt_fwd_grid_mult.load());
        grid = (grid + 7) & ~7;  // whole blocks per XCD residue
        const size_t lds = (size_t)uwpb * 8 * (2 * L * P + 1) * 16;
#define MSDA_LAUNCH_FWD(PTS)                                                                                         \
    hipLaunchKernelGGL(msda_fwd_d32_gather<PTS>, dim3(grid), dim3(use_block), lds, stream, (const float *)value,     \
                       shapes, lstart, (const float *)loc, (const float *)attn, N, S, M, L, Lq, P, (float *)out,     \
                       (unsigned)value_bytes)
        if (variant == 3) {
            g_kernel = "msda_fwd_d32_gather<4>";
            MSDA_LAUNCH_FWD(4);
        } else if (variant == 4) {
            g_kernel = "msda_fwd_d32_gather<1>";
            MSDA_LAUNCH_FWD(1);
        } else {
            g_kernel = "msda_fwd_d32_gather<2>";
            MSDA_LAUNCH_FWD(2);
        }
#undef MSDA_LAUNCH_FWD
        return check_launch(g_kernel);
    }
    return fail(MSDA_ENOTSUP, "no specialised forward for this dtype");
}

template <typename TV, typename TC, typename TG>
int backward_impl(const TV *value, const int64_t *shapes, const int64_t *lstart, const TC *loc, const TC *attn,
                  const TV *grad_out, int N, int S, int M, int D, int L, int Lq, int P, TG *grad_value, TC *grad_loc,
                  TC *grad_attn, int zero_grad_value, hipStream_t stream, bool allow_d32) {
    int rc = check_dims(value, shapes, lstart, loc, attn, grad_out, N, S, M, D, L, Lq, P);
    if (rc) return rc;
    if (!grad_value || !grad_loc || !grad_attn) return fail(MSDA_EINVAL, "null gradient pointer");
    if (zero_grad_value) {
        const hipError_t e = hipMemsetAsync(grad_value, 0, (size_t)N * S * M * D * sizeof(TG), stream);
        if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    }
    if ((long)N * Lq == 0) { g_err[0] = 0; return MSDA_OK; }
    int variant = opt_bwd_variant.load();
    const long value_bytes = (long)N * S * M * D * (long)sizeof(TV);
    const bool can32 = allow_d32 && d32_ok(D, L, value_bytes);
    if (variant == 0) variant = can32 ? 2 : 1;
    if (variant >= 2 && !can32) variant = 1;
    if (variant == 1) {
        int block = ((D + 63) / 64) * 64;
        if (block > 1024) block = 1024;
        const long n_rows = (long)N * Lq * M;
        const int grid = (int)(n_rows < 65536L * 16 ? n_rows : 65536L * 16);
        g_kernel = "msda_bwd_generic";
        hipLaunchKernelGGL((msda_bwd_generic<TV, TC, TG>), dim3(grid), dim3(block), 0, stream, value, shapes, lstart,
                           loc, attn, grad_out, N, S, M, D, L, Lq, P, grad_value, grad_loc, grad_attn);
        return check_launch("msda_bwd_generic");
    }
    if constexpr (sizeof(TV) == 4 && sizeof(TC) == 4 && sizeof(TG) == 4) {
        int block = opt_bwd_block.load();
        if (block < 64 || block > 1024 || (block & 63)) block = 256;
        const int wpb = block / 64;
        const long n_tasks = ((long)N * Lq * M + 7) / 8;
        int use_block = block;
        if (n_tasks < (long)kNumCU * wpb) use_block = 64;
        const int uwpb = use_block / 64;
        int grid = clamp_grid((n_tasks + uwpb - 1) / uwpb, opt_bwd_grid_mult.load());
        grid = (grid + 7) & ~7;
        const size_t lds = (size_t)uwpb * 8 * (2 * L * P + 1) * 16 + (size_t)uwpb * 8 * (3 * L * P + 1) * 4;
#define MSDA_LAUNCH_BWD(PTS, ATOM)                                                                                   \
    hipLaunchKernelGGL((msda_bwd_d32_gather<PTS, ATOM>), dim3(grid), dim3(use_block), lds, stream,                   \
                       (const float *)value, shapes, lstart, (const float *)loc, (const float *)attn,                \
                       (const float *)grad_out, N, S, M, L, Lq, P, (float *)grad_value, (float *)grad_loc,           \
                       (float *)grad_attn, (unsigned)value_bytes)
        if (variant == 3) {
            g_kernel = "msda_bwd_d32_gather<2>";
            MSDA_LAUNCH_BWD(2, true);
        } else if (variant == 90) {
            g_kernel = "msda_bwd_d32_gather<1,noatomics>";
            MSDA_LAUNCH_BWD(1, false);
        } else {
            g_kernel = "msda_bwd_d32_gather<1>";
            MSDA_LAUNCH_BWD(1, true);
        }
#undef MSDA_LAUNCH_BWD
        return check_launch(g_kernel);
    }
    return fail(MSDA_ENOTSUP, "no specialised backward for this dtype");
}

}  // namespace

extern "C" {

int msda_abi_version(void) { return 1; }
const char *msda_last_error(void) { return g_err; }
const char *msda_last_kernel(void) { return g_kernel; }

int msda_forward_f32(const float *value, const int64_t *shapes_dev, const int64_t *lstart_dev, const float *loc,
                     const float *attn, int N, int S, int M, int D, int L, int Lq, int P, float *out,
                     const int64_t *shapes_host, void *stream) {
    (void)shapes_host;
    return forward_impl<float, float>(value, shapes_dev, lstart_dev, loc, attn, N, S, M, D, L, Lq, P, out,
                                      (hipStream_t)stream, true);
}

int msda_forward_f64(const double *value, const int64_t *shapes_dev, const int64_t *lstart_dev, const double *loc,
                     const double *attn, int N, int S, int M, int D, int L, int Lq, int P, double *out,
                     const int64_t *shapes_host, void *stream) {
    (void)shapes_host;
    return forward_impl<double, double>(value, shapes_dev, lstart_dev, loc, attn, N, S, M, D, L, Lq, P, out,
                                        (hipStream_t)stream, false);
}

int msda_forward_bf16(const uint16_t *value, const int64_t *shapes_dev, const int64_t *lstart_dev, const float *loc,
                      const float *attn, int N, int S, int M, int D, int L, int Lq, int P, uint16_t *out,
                      const int64_t *shapes_host, void *stream) {
    (void)shapes_host;
    return forward_impl<bf16_t, float>((const bf16_t *)value, shapes_dev, lstart_dev, loc, attn, N, S, M, D, L, Lq, P,
                                       (bf16_t *)out, (hipStream_t)stream, false);
}

int msda_backward_f32(const float *value, const int64_t *shapes_dev, const int64_t *lstart_dev, const float *loc,
                      const float *attn, const float *grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                      float *grad_value, float *grad_loc, float *grad_attn, int zero_grad_value,
                      const int64_t *shapes_host, void *stream) {
    (void)shapes_host;
    return backward_impl<float, float, float>(value, shapes_dev, lstart_dev, loc, attn, grad_out, N, S, M, D, L, Lq, P,
                                              grad_value, grad_loc, grad_attn, zero_grad_value, (hipStream_t)stream,
                                              true);
}

int msda_backward_f64(const double *value, const int64_t *shapes_dev, const int64_t *lstart_dev, const double *loc,
                      const double *attn, const double *grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                      double *grad_value, double *grad_loc, double *grad_attn, int zero_grad_value,
                      const int64_t *shapes_host, void *stream) {
    (void)shapes_host;
    return backward_impl<double, double, double>(value, shapes_dev, lstart_dev, loc, attn, grad_out, N, S, M, D, L, Lq,
                                                 P, grad_value, grad_loc, grad_attn, zero_grad_value,
                                                 (hipStream_t)stream, false);
}

int msda_backward_bf16(const uint16_t *value, const int64_t *shapes_dev, const int64_t *lstart_dev, const float *loc,
                       const float *attn, const uint16_t *grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                       float *grad_value, float *grad_loc, float *grad_attn, int zero_grad_value,
                       const int64_t *shapes_host, void *stream) {
    (void)shapes_host;
    return backward_impl<bf16_t, float, float>((const bf16_t *)value, shapes_dev, lstart_dev, loc, attn,
                                               (const bf16_t *)grad_out, N, S, M, D, L, Lq, P, grad_value, grad_loc,
                                               grad_attn, zero_grad_value, (hipStream_t)stream, false);
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
    return nullptr;
}

int msda_set_option(const char *key, int value) {
    std::atomic<int> *o = find_opt(key);
    if (!o || value < 0) return fail(MSDA_EINVAL, "unknown option or negative value");
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
