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
// Variant numbers (msda_set_option "fwd_variant" / "bwd_variant"; 0 = auto):
//   forward : 1 generic | 2,3,4 d32 gather with 2,4,1 points in flight (3 = default for D = 32 fp32) |
//             5 region-tiled LDS windows | 6,7 gather with the coarsest level resident in LDS
//   backward: 1 generic (default unless tiled applies) | 2,3 d32 gather | 5 tiled, float LDS atomics |
//             6,7 tiled, fixed-point LDS windows (6 = default for pyramid self-attention) | 90 ablation (no scatter)
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

// forward, variant 6: gather with the coarsest pyramid level resident in LDS.
// The last level is tiny (13x21 pixels at 800x1333: 35 KB per head) yet receives 1/L of all corner reads.
// A workgroup serves ONE (batch, head): it copies that head's last-level rows into LDS once and then walks
// groups of 8 consecutive queries (a wavefront owns 8 queries x this head).  Points of the last level read
// their corners with ds_read_b128 (the whole level is resident: no window placement, no fallback); all other
// points take the buffer-load path of msda_fwd_d32_gather.  The level of point t is the same for every row,
// so the LDS/global choice is a wave-uniform branch.
struct GatherLdsPlan {
    int px_last;     // H*W of the last level
    int H_last, W_last;
    int chunks8;     // chunks per XCD residue; a (batch, head) is split into 8*chunks8 chunks
    int gpc;         // query groups (of 8) per chunk
    int Gb;          // query groups per batch element = ceil(Lq / 8)
    int n_blocks;    // N * M * 8 * chunks8
};

template <int PTS>
__global__ __launch_bounds__(256) void msda_fwd_d32_gather_lds(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lstart,
    const float *__restrict__ loc, const float *__restrict__ attn, int N, int S, int M, int L, int Lq, int P,
    float *__restrict__ out, unsigned value_bytes, const GatherLdsPlan gp) {
    constexpr int D = 32;
    __shared__ int s_H[kMaxLevels], s_W[kMaxLevels], s_start[kMaxLevels];
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    if (threadIdx.x < L) {
        s_H[threadIdx.x] = (int)shapes[2 * threadIdx.x];
        s_W[threadIdx.x] = (int)shapes[2 * threadIdx.x + 1];
        s_start[threadIdx.x] = (int)lstart[threadIdx.x];
    }
    // block -> (xcd slab, batch, head, chunk): bid % 8 is the XCD, each XCD owns a contiguous eighth of the queries
    const int bid = blockIdx.x;
    if (bid >= gp.n_blocks) return;
    const int xcd = bid & 7;
    int r = bid >> 3;
    const int m = r % M;
    r /= M;
    const int ci = r % gp.chunks8;
    const int b = r / gp.chunks8;
    const int chunk = xcd * gp.chunks8 + ci;
    const int g_begin = chunk * gp.gpc;
    const int g_end = (g_begin + gp.gpc < gp.Gb) ? g_begin + gp.gpc : gp.Gb;
    __syncthreads();

    const int LP = L * P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 3, sub = lane & 7;
    const __amdgpu_buffer_rsrc_t vr = make_rsrc(value, value_bytes);
    f32x4 *lvl_f4 = reinterpret_cast<f32x4 *>(s_dyn);
    const unsigned zero_row = (unsigned)gp.px_last * 128u;
    const int rec_stride = 2 * LP + 1;
    u32x4 *rec = reinterpret_cast<u32x4 *>(s_dyn + (size_t)(gp.px_last + 1) * 128) + (size_t)(wave * 8 + grp) * rec_stride;
    const unsigned row_base = ((unsigned)b * (unsigned)S * (unsigned)M + (unsigned)m) * (D * 4u);
    const unsigned pix_stride = (unsigned)M * (D * 4u);

    // ---- resident level: coalesced 128-byte rows of this head ----
    if (g_begin < g_end) {
        const unsigned lvl_off = row_base + (unsigned)s_start[L - 1] * pix_stride;
        const int n = gp.px_last * 8;
        for (int i0 = threadIdx.x; i0 < n; i0 += 256 * 4) {
            f32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + j * 256;
                v[j] = buf_load_f4(vr, i < n ? lvl_off + (unsigned)(i >> 3) * pix_stride + (unsigned)(i & 7) * 16u
                                             : kOobOffset);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + j * 256;
                if (i < n) lvl_f4[i] = v[j];
            }
        }
        if (threadIdx.x < 8) lvl_f4[gp.px_last * 8 + threadIdx.x] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();

    const unsigned lane_off = (unsigned)sub * 16u;
    const int last = L - 1;
    for (int g = g_begin + wave; g < g_end; g += 4) {
        const int q = g * 8 + grp;
        const bool row_ok = q < Lq;
        const long pm = (((long)b * Lq + (row_ok ? q : Lq - 1)) * M + m);
        // ---- stage: each lane prepares LP/8 records of its row ----
        for (int t = sub; t < LP; t += 8) {
            const int l = t / P;
            const f32x2 xy = *reinterpret_cast<const f32x2 *>(loc + (pm * LP + t) * 2);
            const float a = attn[pm * LP + t];
            const int H = s_H[l], W = s_W[l];
            const Sample<float> s = sample_setup<float>(xy.x, xy.y, H, W);
            const float hh = 1.f - s.lh, hw = 1.f - s.lw;
            const int h0 = s.h_low, w0 = s.w_low, h1 = h0 + 1, w1 = w0 + 1;
            const bool live = s.gate && row_ok;
            const bool okh0 = live && h0 >= 0, okh1 = live && h1 <= H - 1;
            const bool okw0 = w0 >= 0, okw1 = w1 <= W - 1;
            u32x4 off;
            if (l == last) {   // LDS byte offsets inside the resident level
                const unsigned c00 = (unsigned)(h0 * W + w0) * 128u;
                off.x = (okh0 && okw0) ? c00 : zero_row;
                off.y = (okh0 && okw1) ? c00 + 128u : zero_row;
                off.z = (okh1 && okw0) ? c00 + (unsigned)W * 128u : zero_row;
                off.w = (okh1 && okw1) ? c00 + (unsigned)W * 128u + 128u : zero_row;
            } else {
                const unsigned o00 = row_base + (unsigned)(s_start[l] + h0 * W + w0) * pix_stride;
                off.x = (okh0 && okw0) ? o00 : kOobOffset;
                off.y = (okh0 && okw1) ? o00 + pix_stride : kOobOffset;
                off.z = (okh1 && okw0) ? o00 + (unsigned)W * pix_stride : kOobOffset;
                off.w = (okh1 && okw1) ? o00 + (unsigned)W * pix_stride + pix_stride : kOobOffset;
            }
            f32x4 w;
            w.x = (hh * hw) * a;
            w.y = (hh * s.lw) * a;
            w.z = (s.lh * hw) * a;
            w.w = (s.lh * s.lw) * a;
            rec[2 * t] = off;
            rec[2 * t + 1] = __builtin_bit_cast(u32x4, w);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int t0 = 0; t0 < LP; t0 += PTS) {
            u32x4 o[PTS];
            f32x4 w[PTS], v[PTS][4];
#pragma unroll
            for (int i = 0; i < PTS; ++i) {
                const int t = t0 + i < LP ? t0 + i : LP - 1;
                o[i] = rec[2 * t];
                w[i] = __builtin_bit_cast(f32x4, rec[2 * t + 1]);
                if (t0 + i >= LP) w[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int i = 0; i < PTS; ++i) {
                const int t = t0 + i < LP ? t0 + i : LP - 1;
                if (t / P == last) {     // wave-uniform
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        v[i][k] = *reinterpret_cast<const f32x4 *>(s_dyn + (o[i][k] + lane_off));
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[i][k] = buf_load_f4(vr, o[i][k] + lane_off);
                }
            }
#pragma unroll
            for (int i = 0; i < PTS; ++i) {
                acc += w[i].x * v[i][0];
                acc += w[i].y * v[i][1];
                acc += w[i].z * v[i][2];
                acc += w[i].w * v[i][3];
            }
        }
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
// Region-tiled kernels for self-attention over the pyramid (one query per pixel, Lq == S).
//
// A workgroup owns one (batch, region, head).  A region is the set of queries whose pixels fall
// in one cell of the coarsest level's grid: side_l = 2^(L-1-l) pixels per side at level l
// (8x8 + 4x4 + 2x2 + 1 = 85 queries for L = 4).  All of them sample the same neighbourhood of
// every level, so the workgroup keeps one window of this head's rows per level in LDS
// (128 B per pixel), centred on the mean sampling position it measures first:
//   forward : windows hold `value`; corner reads are ds_read_b128 instead of L1/L2 requests
//   backward: windows hold the grad_value partial sums; corner scatters are ds_add_f32 and the
//             windows are flushed once with coalesced global atomics
// Corners that fall outside a window take the global path (buffer load / buffer atomic), so
// results do not depend on where the samples are -- only the speed does.
// ----------------------------------------------------------------------------------------
constexpr int kTileMaxL = 4;
constexpr int kTileThreads = 256;
constexpr unsigned kGlobalTag = 0x80000000u;

struct TilePlan {
    int N, S, M, L, P, Lq;
    int RY, RX;
    int rows;                      // queries per region
    int H[kTileMaxL], W[kTileMaxL];
    int qstart[kTileMaxL];         // first query of level l (cumulative H*W)
    int shift[kTileMaxL];          // log2(side_l)
    int row0[kTileMaxL + 1];       // first region-row of level l
    int win[kTileMaxL];            // window side in pixels
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
    long pm;
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
    const int q = o.ok ? tb.qstart[l] + py * tb.W[l] + px : 0;
    o.pm = ((long)b * Lq + q) * M + m;
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
    }
    if (t <= kTileMaxL) {
        tb.row0[t] = pl.row0[t];
        tb.base[t] = pl.win_base[t];
    }
}

// Measure the mean sampling position of every level over the region's gated points and place
// the windows around it.  Ends with a __syncthreads(); tb.oy/ox are valid afterwards.
__device__ __forceinline__ void tile_place_windows(TileTables &tb, const TilePlan &pl, const float *__restrict__ loc,
                                                   int b, int ry, int rx, int m) {
    const int LP = pl.L * pl.P;
    const int lane = threadIdx.x & 63;
    for (int l = 0; l < pl.L; ++l) {
        float sx = 0.f, sy = 0.f, cnt = 0.f;
        const int n = pl.rows * pl.P;
        const int H = tb.H[l], W = tb.W[l];
        for (int i = threadIdx.x; i < n; i += kTileThreads) {
            const int r = i / pl.P, p = i - r * pl.P;
            const TileRow row = tile_row(tb, pl.L, pl.rows, r, b, ry, rx, m, pl.M, pl.Lq);
            if (row.ok) {
                const f32x2 xy = *reinterpret_cast<const f32x2 *>(loc + (row.pm * LP + l * pl.P + p) * 2);
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
    if (threadIdx.x < pl.L) {
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

// Byte offset of pixel (gy, gx) of level l, head m, batch b, relative to the tensor base.
__device__ __forceinline__ unsigned tile_pixel_off(const TileTables &tb, const TilePlan &pl, int b, int l, int gy,
                                                   int gx, int m) {
    return (((unsigned)b * (unsigned)pl.S + (unsigned)(tb.lstart[l] + gy * tb.W[l] + gx)) * (unsigned)pl.M +
            (unsigned)m) * 128u;
}

// LDS byte offset of a window cell, or a tagged global offset when the corner is outside the window;
// `dead` when the corner is outside the level (or the point is gated off).
__device__ __forceinline__ unsigned tile_corner_target(const TileTables &tb, const TilePlan &pl, int b, int l, int cy,
                                                       int cx, int m, bool valid, unsigned dead) {
    if (!valid) return dead;
    const int wy = cy - tb.oy[l], wx = cx - tb.ox[l], win = tb.win[l];
    if ((unsigned)wy < (unsigned)win && (unsigned)wx < (unsigned)win)
        return (unsigned)(tb.base[l] + wy * win + wx) * 128u;
    return kGlobalTag | tile_pixel_off(tb, pl, b, l, cy, cx, m);
}

// ---- forward -------------------------------------------------------------------------------
__global__ __launch_bounds__(kTileThreads) void msda_fwd_d32_tile(
    const float *__restrict__ value, const int64_t *__restrict__ lstart, const float *__restrict__ loc,
    const float *__restrict__ attn, float *__restrict__ out, const TilePlan pl) {
    constexpr int D = 32;
    __shared__ TileTables tb;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    int b, ry, rx, m;
    bool live;
    tile_block_coords(pl, b, ry, rx, m, live);
    if (!live) return;
    tile_load_tables(tb, pl, lstart);
    __syncthreads();
    tile_place_windows(tb, pl, loc, b, ry, rx, m);

    const int LP = pl.L * pl.P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 3, sub = lane & 7;
    const __amdgpu_buffer_rsrc_t vr = make_rsrc(value, pl.value_bytes);
    f32x4 *win_f4 = reinterpret_cast<f32x4 *>(s_dyn);
    const int win_px = tb.base[pl.L];
    const unsigned zero_row = (unsigned)win_px * 128u;   // one all-zero pixel row after the windows
    const int rec_stride = 2 * LP + 1;
    u32x4 *rec = reinterpret_cast<u32x4 *>(s_dyn + (size_t)(win_px + 1) * 128) + (size_t)(wave * 8 + grp) * rec_stride;

    // ---- fill the windows (coalesced 128-byte rows; out-of-level cells read as zero) ----
    if (threadIdx.x < 8) win_f4[win_px * 8 + threadIdx.x] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < pl.L; ++l) {
        const int win = tb.win[l], magic = tb.magic[l], base = tb.base[l];
        const int oy = tb.oy[l], ox = tb.ox[l], H = tb.H[l], W = tb.W[l];
        const int n = win * win * 8;
#pragma unroll 4
        for (int i = threadIdx.x; i < n; i += kTileThreads) {
            const int pix = i >> 3, s8 = i & 7;
            const int wy = (pix * magic) >> 16, wx = pix - wy * win;
            const int gy = oy + wy, gx = ox + wx;
            const unsigned off = (gy < H && gx < W) ? tile_pixel_off(tb, pl, b, l, gy, gx, m) + (unsigned)s8 * 16u
                                                    : kOobOffset;
            win_f4[(base + pix) * 8 + s8] = buf_load_f4(vr, off);
        }
    }
    __syncthreads();

    // ---- passes of 32 rows (8 per wavefront) ----
    const unsigned lane_off = (unsigned)sub * 16u;
    for (int r0 = 0; r0 < pl.rows; r0 += 32) {
        const TileRow row = tile_row(tb, pl.L, pl.rows, r0 + wave * 8 + grp, b, ry, rx, m, pl.M, pl.Lq);
        for (int t = sub; t < LP; t += 8) {
            const int l = t / pl.P;
            const f32x2 xy = *reinterpret_cast<const f32x2 *>(loc + (row.pm * LP + t) * 2);
            const float a = attn[row.pm * LP + t];
            const int H = tb.H[l], W = tb.W[l];
            const Sample<float> s = sample_setup<float>(xy.x, xy.y, H, W);
            const float hh = 1.f - s.lh, hw = 1.f - s.lw;
            const int h0 = s.h_low, w0 = s.w_low, h1 = h0 + 1, w1 = w0 + 1;
            const bool livep = s.gate && row.ok;
            const bool okh0 = livep && h0 >= 0, okh1 = livep && h1 <= H - 1;
            const bool okw0 = w0 >= 0, okw1 = w1 <= W - 1;
            u32x4 off;
            off.x = tile_corner_target(tb, pl, b, l, h0, w0, m, okh0 && okw0, zero_row);
            off.y = tile_corner_target(tb, pl, b, l, h0, w1, m, okh0 && okw1, zero_row);
            off.z = tile_corner_target(tb, pl, b, l, h1, w0, m, okh1 && okw0, zero_row);
            off.w = tile_corner_target(tb, pl, b, l, h1, w1, m, okh1 && okw1, zero_row);
            f32x4 w;
            w.x = (hh * hw) * a;
            w.y = (hh * s.lw) * a;
            w.z = (s.lh * hw) * a;
            w.w = (s.lh * s.lw) * a;
            rec[2 * t] = off;
            rec[2 * t + 1] = __builtin_bit_cast(u32x4, w);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int t = 0; t < LP; ++t) {
            const u32x4 o = rec[2 * t];
            const f32x4 w = __builtin_bit_cast(f32x4, rec[2 * t + 1]);
            f32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned ok = o[k];
                const bool g = (ok & kGlobalTag) != 0u;
                v[k] = *reinterpret_cast<const f32x4 *>(s_dyn + ((g ? zero_row : ok) + lane_off));
                if (g) v[k] = buf_load_f4(vr, (ok & ~kGlobalTag) + lane_off);
            }
            acc += w.x * v[0];
            acc += w.y * v[1];
            acc += w.z * v[2];
            acc += w.w * v[3];
        }
        if (row.ok) *reinterpret_cast<f32x4 *>(out + row.pm * D + sub * 4) = acc;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- backward ------------------------------------------------------------------------------
// Records: [0] global value offsets of the 4 corners (kOobOffset when dead), [1] scatter targets
// (LDS window byte offset, or kGlobalTag|global offset, dead -> tagged out-of-range), [2] lh, lw, attn.
__global__ __launch_bounds__(kTileThreads) void msda_bwd_d32_tile(
    const float *__restrict__ value, const int64_t *__restrict__ lstart, const float *__restrict__ loc,
    const float *__restrict__ attn, const float *__restrict__ grad_out, float *__restrict__ grad_value,
    float *__restrict__ grad_loc, float *__restrict__ grad_attn, const TilePlan pl) {
    constexpr int D = 32;
    __shared__ TileTables tb;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    int b, ry, rx, m;
    bool live;
    tile_block_coords(pl, b, ry, rx, m, live);
    if (!live) return;
    tile_load_tables(tb, pl, lstart);
    __syncthreads();

    const int LP = pl.L * pl.P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 3, sub = lane & 7;
    const __amdgpu_buffer_rsrc_t vr = make_rsrc(value, pl.value_bytes);
    const __amdgpu_buffer_rsrc_t gr = make_rsrc(grad_value, pl.value_bytes);
    f32x4 *win_f4 = reinterpret_cast<f32x4 *>(s_dyn);
    float *win_f = reinterpret_cast<float *>(s_dyn);
    const int win_px = tb.base[pl.L];
    const int rec_stride = 3 * LP + 1;
    u32x4 *rec = reinterpret_cast<u32x4 *>(s_dyn + (size_t)win_px * 128) + (size_t)(wave * 8 + grp) * rec_stride;
    float *res = reinterpret_cast<float *>(s_dyn + (size_t)win_px * 128 + (size_t)32 * rec_stride * 16) +
                 (size_t)(wave * 8 + grp) * (3 * LP + 1);

    // zero the accumulation windows while the placement pass runs
    for (int i = threadIdx.x; i < win_px * 8; i += kTileThreads) win_f4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    tile_place_windows(tb, pl, loc, b, ry, rx, m);  // ends with __syncthreads()

    const unsigned lane_off = (unsigned)sub * 16u;
    const unsigned dead_target = kGlobalTag | 0x7fffff00u;  // tagged global, beyond value_bytes -> dropped
    for (int r0 = 0; r0 < pl.rows; r0 += 32) {
        const TileRow row = tile_row(tb, pl.L, pl.rows, r0 + wave * 8 + grp, b, ry, rx, m, pl.M, pl.Lq);
        for (int t = sub; t < LP; t += 8) {
            const int l = t / pl.P;
            const f32x2 xy = *reinterpret_cast<const f32x2 *>(loc + (row.pm * LP + t) * 2);
            const float a = row.ok ? attn[row.pm * LP + t] : 0.f;
            const int H = tb.H[l], W = tb.W[l];
            const Sample<float> s = sample_setup<float>(xy.x, xy.y, H, W);
            const int h0 = s.h_low, w0 = s.w_low, h1 = h0 + 1, w1 = w0 + 1;
            const bool livep = s.gate && row.ok;
            const bool okh0 = livep && h0 >= 0, okh1 = livep && h1 <= H - 1;
            const bool okw0 = w0 >= 0, okw1 = w1 <= W - 1;
            const unsigned ps = (unsigned)pl.M * 128u;
            const unsigned o00 = tile_pixel_off(tb, pl, b, l, h0, w0, m);
            u32x4 go, to;
            go.x = (okh0 && okw0) ? o00 : kOobOffset;
            go.y = (okh0 && okw1) ? o00 + ps : kOobOffset;
            go.z = (okh1 && okw0) ? o00 + (unsigned)W * ps : kOobOffset;
            go.w = (okh1 && okw1) ? o00 + (unsigned)W * ps + ps : kOobOffset;
            to.x = tile_corner_target(tb, pl, b, l, h0, w0, m, okh0 && okw0, dead_target);
            to.y = tile_corner_target(tb, pl, b, l, h0, w1, m, okh0 && okw1, dead_target);
            to.z = tile_corner_target(tb, pl, b, l, h1, w0, m, okh1 && okw0, dead_target);
            to.w = tile_corner_target(tb, pl, b, l, h1, w1, m, okh1 && okw1, dead_target);
            f32x4 w;
            w.x = s.lh;
            w.y = s.lw;
            w.z = a;
            w.w = 0.f;
            rec[3 * t] = go;
            rec[3 * t + 1] = to;
            rec[3 * t + 2] = __builtin_bit_cast(u32x4, w);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const f32x4 g = row.ok ? *reinterpret_cast<const f32x4 *>(grad_out + row.pm * D + sub * 4)
                               : f32x4{0.f, 0.f, 0.f, 0.f};
        // LDS atomics: rotate the channel order by the row slot so that the 64 lanes of one
        // ds_add_f32 cover all 32 banks twice (rows are 128-byte aligned: without the rotation
        // eight rows would pile onto the same eight banks).
        const int rot = grp & 3;
        const f32x4 g_rot = rot == 0 ? g : rot == 1 ? f32x4{g.y, g.z, g.w, g.x}
                                  : rot == 2 ? f32x4{g.z, g.w, g.x, g.y} : f32x4{g.w, g.x, g.y, g.z};
        const unsigned c0 = (unsigned)((0 + rot) & 3) * 4u, c1 = (unsigned)((1 + rot) & 3) * 4u;
        const unsigned c2 = (unsigned)((2 + rot) & 3) * 4u, c3 = (unsigned)((3 + rot) & 3) * 4u;
        for (int t = 0; t < LP; ++t) {
            const u32x4 go = rec[3 * t];
            const u32x4 to = rec[3 * t + 1];
            const f32x4 rw = __builtin_bit_cast(f32x4, rec[3 * t + 2]);
            const float lh = rw.x, lw = rw.y, a = rw.z;
            const float hh = 1.f - lh, hw = 1.f - lw;
            const f32x4 v1 = buf_load_f4(vr, go.x + lane_off);
            const f32x4 v2 = buf_load_f4(vr, go.y + lane_off);
            const f32x4 v3 = buf_load_f4(vr, go.z + lane_off);
            const f32x4 v4 = buf_load_f4(vr, go.w + lane_off);
            const f32x4 tga = g * a;
            const f32x4 tga_rot = g_rot * a;
            const float wk[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned tk = to[k];
                const f32x4 c = wk[k] * tga_rot;
                if (tk & kGlobalTag) {
                    const unsigned o = (tk & ~kGlobalTag) + lane_off;
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(c.x, gr, (int)(o + c0), 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(c.y, gr, (int)(o + c1), 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(c.z, gr, (int)(o + c2), 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(c.w, gr, (int)(o + c3), 0, 0);
                } else {
                    unsigned char *p = s_dyn + tk + lane_off;
                    atomicAdd(reinterpret_cast<float *>(p + c0), c.x);
                    atomicAdd(reinterpret_cast<float *>(p + c1), c.y);
                    atomicAdd(reinterpret_cast<float *>(p + c2), c.z);
                    atomicAdd(reinterpret_cast<float *>(p + c3), c.w);
                }
            }
            const f32x4 val = wk[0] * v1 + wk[1] * v2 + wk[2] * v3 + wk[3] * v4;
            const f32x4 gw = hh * (v2 - v1) + lh * (v4 - v3);
            const f32x4 gh = hw * (v3 - v1) + lw * (v4 - v2);
            float pa = g.x * val.x + g.y * val.y + g.z * val.z + g.w * val.w;
            float pw = gw.x * tga.x + gw.y * tga.y + gw.z * tga.z + gw.w * tga.w;
            float ph = gh.x * tga.x + gh.y * tga.y + gh.z * tga.z + gh.w * tga.w;
            pa = sum8(pa);
            pw = sum8(pw);
            ph = sum8(ph);
            if (sub == (t & 7)) {
                const int l = t / pl.P;
                res[2 * t] = pw * (float)tb.W[l];
                res[2 * t + 1] = ph * (float)tb.H[l];
                res[2 * LP + t] = pa;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (row.ok) {
            for (int i = sub; i < 2 * LP; i += 8) grad_loc[row.pm * LP * 2 + i] = res[i];
            for (int i = sub; i < LP; i += 8) grad_attn[row.pm * LP + i] = res[2 * LP + i];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // ---- flush: one coalesced global atomic per touched window element ----
    for (int l = 0; l < pl.L; ++l) {
        const int win = tb.win[l], magic = tb.magic[l], base = tb.base[l];
        const int oy = tb.oy[l], ox = tb.ox[l], H = tb.H[l], W = tb.W[l];
        const int n = win * win * D;
        for (int i = threadIdx.x; i < n; i += kTileThreads) {
            const int pix = i >> 5, c = i & 31;
            const float v = win_f[(base + pix) * D + c];
            if (v != 0.f) {
                const int wy = (pix * magic) >> 16, wx = pix - wy * win;
                const int gy = oy + wy, gx = ox + wx;
                if (gy < H && gx < W)
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(
                        v, gr, (int)(tile_pixel_off(tb, pl, b, l, gy, gx, m) + (unsigned)c * 4u), 0, 0);
            }
        }
    }
}

// ---- backward, fixed-point window accumulation ---------------------------------------------------
// Same tiling as msda_bwd_d32_tile, but the LDS windows accumulate grad_value as 32-bit fixed point:
// on gfx950 ds_add_f32 retires ~0.33 lanes/clk/CU while ds_add_u32 retires 5-13 (profiles/r01_ubench_*).
// Per workgroup: B = max|grad_out| * max|attn| over the region bounds every contribution
// (|w_corner| <= 1); a window cell receives at most rows*P contributions (one per point of its
// level), so with scale = 2^e, e = 30 - ceil(log2(rows*P)) - ceil(log2(B)), sums cannot overflow and
// the quantisation step is B * 2^-(30 - log2(rows*P)) (~2^-21 B for L = P = 4).  The flush converts
// back (exact power-of-two scaling) and adds into grad_value with float atomics like every other
// path.  Integer adds commute, so the in-window part of the result is order-independent.
//
// Records are 32 bytes per (row, point): {global offset of corner 00, window cells 00|01, 10|11
// (0xffff = outside the window), valid mask} + {lh, lw, attn}.  For the scatter each lane owns
// channels {sub, sub+8, sub+16, sub+24} (its own strided copy of grad_out): the 8 lanes of a row then
// write 32 contiguous bytes per atomic -- twice the L2 atomic rate of the 16-byte-strided float4
// layout on the fallback path -- and rotating the channel group by the row slot spreads one ds_add
// over all 32 LDS banks.
constexpr unsigned kNoCell = 0xffffu;

template <int PTS>
__global__ __launch_bounds__(kTileThreads) void msda_bwd_d32_tile_q(
    const float *__restrict__ value, const int64_t *__restrict__ lstart, const float *__restrict__ loc,
    const float *__restrict__ attn, const float *__restrict__ grad_out, float *__restrict__ grad_value,
    float *__restrict__ grad_loc, float *__restrict__ grad_attn, const TilePlan pl) {
    constexpr int D = 32;
    __shared__ TileTables tb;
    __shared__ int s_bound[2];   // float bits of max|grad_out|, max|attn|
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    int b, ry, rx, m;
    bool live;
    tile_block_coords(pl, b, ry, rx, m, live);
    if (!live) return;
    tile_load_tables(tb, pl, lstart);
    if (threadIdx.x < 2) s_bound[threadIdx.x] = 0;
    __syncthreads();

    const int LP = pl.L * pl.P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 3, sub = lane & 7;
    const __amdgpu_buffer_rsrc_t vr = make_rsrc(value, pl.value_bytes);
    const __amdgpu_buffer_rsrc_t gr = make_rsrc(grad_value, pl.value_bytes);
    u32x4 *win_u4 = reinterpret_cast<u32x4 *>(s_dyn);
    int *win_i = reinterpret_cast<int *>(s_dyn);
    const int win_px = tb.base[pl.L];
    const int rec_stride = 2 * LP + 1;
    // layout: [windows | 8 dump rows (one per row slot: dead / out-of-window corners add zeros there, so the
    // in-window scatter needs no branches) | records | results]
    const unsigned dump_cell = (unsigned)(win_px + grp);
    u32x4 *rec = reinterpret_cast<u32x4 *>(s_dyn + (size_t)(win_px + 8) * 128) + (size_t)(wave * 8 + grp) * rec_stride;
    float *res = reinterpret_cast<float *>(s_dyn + (size_t)(win_px + 8) * 128 + (size_t)32 * rec_stride * 16) +
                 (size_t)(wave * 8 + grp) * (3 * LP + 1);

    for (int i = threadIdx.x; i < (win_px + 8) * 8; i += kTileThreads) win_u4[i] = u32x4{0u, 0u, 0u, 0u};
    {   // contribution bound of this region: max|grad_out| and max|attn| over its rows
        float gmax = 0.f, amax = 0.f;
        for (int i = threadIdx.x; i < pl.rows * 8; i += kTileThreads) {
            const TileRow row = tile_row(tb, pl.L, pl.rows, i >> 3, b, ry, rx, m, pl.M, pl.Lq);
            if (row.ok) {
                const f32x4 g4 = *reinterpret_cast<const f32x4 *>(grad_out + row.pm * D + (i & 7) * 4);
                gmax = fmaxf(gmax, fmaxf(fmaxf(fabsf(g4.x), fabsf(g4.y)), fmaxf(fabsf(g4.z), fabsf(g4.w))));
            }
        }
        for (int i = threadIdx.x; i < pl.rows * LP; i += kTileThreads) {
            const int r = i / LP;
            const TileRow row = tile_row(tb, pl.L, pl.rows, r, b, ry, rx, m, pl.M, pl.Lq);
            if (row.ok) amax = fmaxf(amax, fabsf(attn[row.pm * LP + (i - r * LP)]));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            gmax = fmaxf(gmax, __shfl_xor(gmax, o, kWave));
            amax = fmaxf(amax, __shfl_xor(amax, o, kWave));
        }
        if (lane == 0) {   // non-negative floats order like their bit patterns
            atomicMax(&s_bound[0], __float_as_int(gmax));
            atomicMax(&s_bound[1], __float_as_int(amax));
        }
    }
    tile_place_windows(tb, pl, loc, b, ry, rx, m);  // ends with __syncthreads()

    float bound = __int_as_float(s_bound[0]) * __int_as_float(s_bound[1]);
    if (!(bound < 3.0e38f)) bound = 3.0e38f;         // inf / nan inputs: results are garbage either way
    int cnt_log2 = 0;
    while ((1 << cnt_log2) < pl.rows * pl.P) ++cnt_log2;
    int bexp = 0;
    (void)frexpf(bound, &bexp);                       // bound <= 2^bexp
    const int e = 30 - cnt_log2 - bexp;
    const float scale = bound > 0.f ? ldexpf(1.0f, e) : 0.f;
    const float inv_scale = bound > 0.f ? ldexpf(1.0f, -e) : 0.f;

    const unsigned lane_off = (unsigned)sub * 16u;
    const unsigned ps = (unsigned)pl.M * 128u;
    const int rot = grp & 3;
    // byte offset (inside a 128-byte pixel row) of the channel this lane scatters in step j
    unsigned ch_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ch_off[j] = (unsigned)(sub + 8 * ((j + rot) & 3)) * 4u;

    for (int r0 = 0; r0 < pl.rows; r0 += 32) {
        const TileRow row = tile_row(tb, pl.L, pl.rows, r0 + wave * 8 + grp, b, ry, rx, m, pl.M, pl.Lq);
        for (int t = sub; t < LP; t += 8) {
            const int l = t / pl.P;
            const f32x2 xy = *reinterpret_cast<const f32x2 *>(loc + (row.pm * LP + t) * 2);
            const float a = row.ok ? attn[row.pm * LP + t] : 0.f;
            const int H = tb.H[l], W = tb.W[l];
            const Sample<float> s = sample_setup<float>(xy.x, xy.y, H, W);
            const int h0 = s.h_low, w0 = s.w_low, h1 = h0 + 1, w1 = w0 + 1;
            const bool livep = s.gate && row.ok;
            const bool okh0 = livep && h0 >= 0, okh1 = livep && h1 <= H - 1;
            const bool okw0 = w0 >= 0, okw1 = w1 <= W - 1;
            const int win = tb.win[l], base = tb.base[l];
            const int wy0 = h0 - tb.oy[l], wx0 = w0 - tb.ox[l];
            const bool iy0 = (unsigned)wy0 < (unsigned)win, iy1 = (unsigned)(wy0 + 1) < (unsigned)win;
            const bool ix0 = (unsigned)wx0 < (unsigned)win, ix1 = (unsigned)(wx0 + 1) < (unsigned)win;
            const unsigned c00 = (unsigned)(base + wy0 * win + wx0);
            u32x4 r0v;
            r0v.x = tile_pixel_off(tb, pl, b, l, h0, w0, m);
            r0v.y = ((iy0 && ix0) ? c00 : kNoCell) | (((iy0 && ix1) ? c00 + 1u : kNoCell) << 16);
            r0v.z = ((iy1 && ix0) ? c00 + (unsigned)win : kNoCell) | (((iy1 && ix1) ? c00 + (unsigned)win + 1u : kNoCell) << 16);
            r0v.w = (unsigned)(okh0 && okw0) | ((unsigned)(okh0 && okw1) << 1) | ((unsigned)(okh1 && okw0) << 2) |
                    ((unsigned)(okh1 && okw1) << 3);
            f32x4 w;
            w.x = s.lh;
            w.y = s.lw;
            w.z = a;
            w.w = 0.f;
            rec[2 * t] = r0v;
            rec[2 * t + 1] = __builtin_bit_cast(u32x4, w);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const f32x4 g = row.ok ? *reinterpret_cast<const f32x4 *>(grad_out + row.pm * D + sub * 4)
                               : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 gs;   // grad_out at the scatter channels of this lane, pre-multiplied by the fixed-point scale
#pragma unroll
        for (int j = 0; j < 4; ++j) gs[j] = row.ok ? grad_out[row.pm * D + (ch_off[j] >> 2)] : 0.f;
        for (int t0 = 0; t0 < LP; t0 += PTS) {
            u32x4 ra[PTS];
            f32x4 rw[PTS], v[PTS][4];
            unsigned go[PTS][4];
#pragma unroll
            for (int i = 0; i < PTS; ++i) {
                const int t = (t0 + i < LP) ? t0 + i : LP - 1;
                ra[i] = rec[2 * t];
                rw[i] = __builtin_bit_cast(f32x4, rec[2 * t + 1]);
                const unsigned wps = (unsigned)tb.W[t / pl.P] * ps;
                const unsigned mask = ra[i].w;
                go[i][0] = (mask & 1u) ? ra[i].x : kOobOffset;
                go[i][1] = (mask & 2u) ? ra[i].x + ps : kOobOffset;
                go[i][2] = (mask & 4u) ? ra[i].x + wps : kOobOffset;
                go[i][3] = (mask & 8u) ? ra[i].x + wps + ps : kOobOffset;
            }
#pragma unroll
            for (int i = 0; i < PTS; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) v[i][k] = buf_load_f4(vr, go[i][k] + lane_off);
#pragma unroll
            for (int i = 0; i < PTS; ++i) {
                const int t = t0 + i;
                if (t < LP) {
                    const float lh = rw[i].x, lw = rw[i].y, a = rw[i].z;
                    const float hh = 1.f - lh, hw = 1.f - lw;
                    const f32x4 tga = g * a;
                    const f32x4 sa = gs * a;            // contributions before the corner weight
                    const f32x4 sa_q = sa * scale;      // ... in fixed-point units
                    const float wk[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
                    const unsigned cells[4] = {ra[i].y & 0xffffu, ra[i].y >> 16, ra[i].z & 0xffffu, ra[i].z >> 16};
                    bool any_fallback = false;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        // branch-free window scatter: dead and out-of-window corners add 0 to this row slot's dump row
                        const bool alive = go[i][k] != kOobOffset;
                        const bool inwin = alive && cells[k] != kNoCell;
                        any_fallback |= alive && !inwin;
                        const float wq = inwin ? wk[k] : 0.f;
                        unsigned char *p = s_dyn + (inwin ? cells[k] : dump_cell) * 128u;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            atomicAdd(reinterpret_cast<int *>(p + ch_off[j]), __float2int_rn(wq * sa_q[j]));
                    }
                    if (__builtin_amdgcn_ballot_w64(any_fallback) != 0ull) {   // rare: corners outside the windows
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if (go[i][k] != kOobOffset && cells[k] == kNoCell) {
                                const f32x4 c = wk[k] * sa;
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(c[j], gr, (int)(go[i][k] + ch_off[j]), 0, 0);
                            }
                        }
                    }
                    const f32x4 val = wk[0] * v[i][0] + wk[1] * v[i][1] + wk[2] * v[i][2] + wk[3] * v[i][3];
                    const f32x4 gw = hh * (v[i][1] - v[i][0]) + lh * (v[i][3] - v[i][2]);
                    const f32x4 gh = hw * (v[i][2] - v[i][0]) + lw * (v[i][3] - v[i][1]);
                    float pa = g.x * val.x + g.y * val.y + g.z * val.z + g.w * val.w;
                    float pw = gw.x * tga.x + gw.y * tga.y + gw.z * tga.z + gw.w * tga.w;
                    float ph = gh.x * tga.x + gh.y * tga.y + gh.z * tga.z + gh.w * tga.w;
                    pa = sum8(pa);
                    pw = sum8(pw);
                    ph = sum8(ph);
                    if (sub == (t & 7)) {
                        const int l = t / pl.P;
                        res[2 * t] = pw * (float)tb.W[l];
                        res[2 * t + 1] = ph * (float)tb.H[l];
                        res[2 * LP + t] = pa;
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (row.ok) {
            for (int i = sub; i < 2 * LP; i += 8) grad_loc[row.pm * LP * 2 + i] = res[i];
            for (int i = sub; i < LP; i += 8) grad_attn[row.pm * LP + i] = res[2 * LP + i];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    for (int l = 0; l < pl.L; ++l) {
        const int win = tb.win[l], magic = tb.magic[l], base = tb.base[l];
        const int oy = tb.oy[l], ox = tb.ox[l], H = tb.H[l], W = tb.W[l];
        const int n = win * win * D;
        for (int i = threadIdx.x; i < n; i += kTileThreads) {
            const int pix = i >> 5, c = i & 31;
            const int q = win_i[(base + pix) * D + c];
            if (q != 0) {
                const int wy = (pix * magic) >> 16, wx = pix - wy * win;
                const int gy = oy + wy, gx = ox + wx;
                if (gy < H && gx < W)
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(
                        (float)q * inv_scale, gr, (int)(tile_pixel_off(tb, pl, b, l, gy, gx, m) + (unsigned)c * 4u), 0, 0);
            }
        }
    }
}

// ----------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------
thread_local char g_err[256] = {0};
thread_local const char *g_kernel = "";

std::atomic<int> opt_fwd_variant{0}, opt_bwd_variant{0};
std::atomic<int> opt_fwd_block{256}, opt_bwd_block{256};
std::atomic<int> opt_fwd_grid_mult{32}, opt_bwd_grid_mult{16};
std::atomic<int> opt_fwd_tile_margin{3}, opt_bwd_tile_margin{3};

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

// Plan the region tiling from the HOST copy of the level shapes.  Returns false when the tiled
// kernels do not apply (then the gather kernels run).
bool make_tile_plan(TilePlan &pl, const int64_t *shapes_host, int N, int S, int M, int D, int L, int Lq, int P,
                    long value_bytes, int margin, size_t extra_lds_per_pixel_row, size_t fixed_lds, size_t &lds) {
    if (!shapes_host || D != 32 || L < 1 || L > kTileMaxL || Lq != S || value_bytes >= 0x7fffff00L) return false;
    if (margin < 0) margin = 0;
    memset(&pl, 0, sizeof(pl));
    pl.N = N; pl.S = S; pl.M = M; pl.L = L; pl.P = P; pl.Lq = Lq;
    pl.value_bytes = (unsigned)value_bytes;
    long q = 0;
    int rows = 0, px = 0, RY = 0, RX = 0;
    for (int l = 0; l < kTileMaxL; ++l) {
        if (l < L) {
            const long H = shapes_host[2 * l], W = shapes_host[2 * l + 1];
            if (H <= 0 || W <= 0 || H > 32767 || W > 32767) return false;
            const int sh = L - 1 - l, side = 1 << sh;
            int win = side + 2 * margin;
            if (win > 32) win = 32;
            pl.H[l] = (int)H; pl.W[l] = (int)W; pl.qstart[l] = (int)q; pl.shift[l] = sh; pl.row0[l] = rows;
            pl.win[l] = win; pl.win_base[l] = px;
            const int magic = 65536 / win + 1;
            for (int x = 0; x < win * win; ++x)
                if (((x * magic) >> 16) != x / win) return false;
            pl.win_magic[l] = magic;
            q += H * W; rows += side * side; px += win * win;
            const int ry = (int)((H + side - 1) / side), rx = (int)((W + side - 1) / side);
            RY = ry > RY ? ry : RY;
            RX = rx > RX ? rx : RX;
        } else {  // inert padding so that table loads stay in range
            pl.H[l] = 1; pl.W[l] = 1; pl.qstart[l] = (int)q; pl.shift[l] = 0; pl.row0[l] = rows;
            pl.win[l] = 1; pl.win_magic[l] = 65537; pl.win_base[l] = px;
        }
    }
    if (q != S) return false;  // host shapes do not describe this value tensor
    pl.row0[kTileMaxL] = rows; pl.win_base[kTileMaxL] = px;
    for (int l = L; l <= kTileMaxL; ++l) { pl.row0[l] = rows; pl.win_base[l] = px; }
    pl.rows = rows; pl.RY = RY; pl.RX = RX;
    const long nb = (long)N * RY * RX * M;
    if (nb > (1L << 30)) return false;
    pl.n_blocks = (int)nb;
    lds = (size_t)px * 128 + fixed_lds + extra_lds_per_pixel_row;
    return lds <= 160 * 1024 - 512;
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
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    done.fetch_or(bit, std::memory_order_release);
    return MSDA_OK;
}

template <typename TV, typename TC>
int forward_impl(const TV *value, const int64_t *shapes, const int64_t *lstart, const TC *loc, const TC *attn, int N,
                 int S, int M, int D, int L, int Lq, int P, TV *out, const int64_t *shapes_host, hipStream_t stream,
                 bool allow_d32) {
    int rc = check_dims(value, shapes, lstart, loc, attn, out, N, S, M, D, L, Lq, P);
    if (rc) return rc;
    if ((long)N * Lq == 0) { g_err[0] = 0; return MSDA_OK; }
    int variant = opt_fwd_variant.load();
    const long value_bytes = (long)N * S * M * D * (long)sizeof(TV);
    const bool can32 = allow_d32 && d32_ok(D, L, value_bytes);
    if (variant == 0) variant = can32 ? 3 : 1;  // 4 points (16 rows) in flight: best of the sweep in profiles/
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
        if (variant == 6 || variant == 7) {
            // gather with the last level resident in LDS: needs the host shapes, a last level of <= 48 KB per head
            // and enough rows to amortise the fill
            const long px_last = shapes_host ? shapes_host[2 * (L - 1)] * shapes_host[2 * (L - 1) + 1] : 0;
            if (shapes_host && L >= 2 && px_last > 0 && px_last * 128 <= 48 * 1024 && (long)N * Lq * M >= 1024) {
                GatherLdsPlan gp;
                gp.px_last = (int)px_last;
                gp.H_last = (int)shapes_host[2 * (L - 1)];
                gp.W_last = (int)shapes_host[2 * (L - 1) + 1];
                gp.Gb = (Lq + 7) / 8;
                int chunks8 = opt_fwd_grid_mult.load() / 4;   // grid_mult 32 -> 8 chunks per XCD -> 64 chunks per (batch, head)
                if (chunks8 < 1) chunks8 = 1;
                while (chunks8 > 1 && (long)8 * chunks8 * 8 > gp.Gb) chunks8 >>= 1;   // at least ~8 groups per chunk
                gp.chunks8 = chunks8;
                gp.gpc = (gp.Gb + 8 * chunks8 - 1) / (8 * chunks8);
                gp.n_blocks = N * M * 8 * chunks8;
                const size_t lds = (size_t)(px_last + 1) * 128 + (size_t)32 * (2 * L * P + 1) * 16;
                if (variant == 7) {
                    g_kernel = "msda_fwd_d32_gather_lds<2>";
                    hipLaunchKernelGGL(msda_fwd_d32_gather_lds<2>, dim3(gp.n_blocks), dim3(256), lds, stream,
                                       (const float *)value, shapes, lstart, (const float *)loc, (const float *)attn, N,
                                       S, M, L, Lq, P, (float *)out, (unsigned)value_bytes, gp);
                } else {
                    g_kernel = "msda_fwd_d32_gather_lds<4>";
                    hipLaunchKernelGGL(msda_fwd_d32_gather_lds<4>, dim3(gp.n_blocks), dim3(256), lds, stream,
                                       (const float *)value, shapes, lstart, (const float *)loc, (const float *)attn, N,
                                       S, M, L, Lq, P, (float *)out, (unsigned)value_bytes, gp);
                }
                return check_launch(g_kernel);
            }
            variant = 3;
        }
        if (variant == 5) {
            TilePlan pl;
            size_t lds = 0;
            const size_t rec_bytes = (size_t)32 * (2 * L * P + 1) * 16;
            if (make_tile_plan(pl, shapes_host, N, S, M, D, L, Lq, P, value_bytes, opt_fwd_tile_margin.load(), 128,
                               rec_bytes, lds)) {
                rc = allow_big_lds(msda_fwd_d32_tile, lds);
                if (rc) return rc;
                const int grid = (pl.n_blocks + 7) & ~7;
                g_kernel = "msda_fwd_d32_tile";
                hipLaunchKernelGGL(msda_fwd_d32_tile, dim3(grid), dim3(kTileThreads), lds, stream,
                                   (const float *)value, lstart, (const float *)loc, (const float *)attn,
                                   (float *)out, pl);
                return check_launch(g_kernel);
            }
            variant = 2;  // tiling does not apply to this call
        }
        int block = opt_fwd_block.load();
        if (block < 64 || block > 256 || (block & 63)) block = 256;  // kernels carry __launch_bounds__(256)
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
                  TC *grad_attn, int zero_grad_value, const int64_t *shapes_host, hipStream_t stream, bool allow_d32) {
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
    // Measured on MI355X (profiles/): per-contribution global float atomics cap the backward at ~1.1 ms
    // for the encoder call (L2 atomic throughput; the row-per-block kernel's 32-consecutive-lane pattern
    // is the fastest of them).  Self-attention over the pyramid (Lq == S, host shapes known) therefore
    // takes the region-tiled kernel that pre-reduces grad_value in fixed-point LDS windows (2x faster on
    // encoder-like sampling); every other call takes the row-per-block kernel.
    if (variant == 0) variant = (can32 && shapes_host && Lq == S && L <= kTileMaxL) ? 6 : 1;
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
        if (variant == 6 || variant == 7) {
            TilePlan pl;
            size_t lds = 0;
            const size_t rec_bytes = (size_t)32 * (2 * L * P + 1) * 16 + (size_t)32 * (3 * L * P + 1) * 4;
            if (make_tile_plan(pl, shapes_host, N, S, M, D, L, Lq, P, value_bytes, opt_bwd_tile_margin.load(), 8 * 128,
                               rec_bytes, lds)) {
                const int grid = (pl.n_blocks + 7) & ~7;
#define MSDA_LAUNCH_TQ(PTS)                                                                                          \
    rc = allow_big_lds(msda_bwd_d32_tile_q<PTS>, lds);                                                               \
    if (rc) return rc;                                                                                               \
    hipLaunchKernelGGL(msda_bwd_d32_tile_q<PTS>, dim3(grid), dim3(kTileThreads), lds, stream, (const float *)value,  \
                       lstart, (const float *)loc, (const float *)attn, (const float *)grad_out,                     \
                       (float *)grad_value, (float *)grad_loc, (float *)grad_attn, pl)
                if (variant == 7) {
                    g_kernel = "msda_bwd_d32_tile_q<8>";
                    MSDA_LAUNCH_TQ(8);
                } else {
                    g_kernel = "msda_bwd_d32_tile_q<4>";
                    MSDA_LAUNCH_TQ(4);
                }
#undef MSDA_LAUNCH_TQ
                return check_launch(g_kernel);
            }
            variant = 1;
        }
        if (variant == 5) {
            TilePlan pl;
            size_t lds = 0;
            const size_t rec_bytes = (size_t)32 * (3 * L * P + 1) * 16 + (size_t)32 * (3 * L * P + 1) * 4;
            if (make_tile_plan(pl, shapes_host, N, S, M, D, L, Lq, P, value_bytes, opt_bwd_tile_margin.load(), 0,
                               rec_bytes, lds)) {
                rc = allow_big_lds(msda_bwd_d32_tile, lds);
                if (rc) return rc;
                const int grid = (pl.n_blocks + 7) & ~7;
                g_kernel = "msda_bwd_d32_tile";
                hipLaunchKernelGGL(msda_bwd_d32_tile, dim3(grid), dim3(kTileThreads), lds, stream,
                                   (const float *)value, lstart, (const float *)loc, (const float *)attn,
                                   (const float *)grad_out, (float *)grad_value, (float *)grad_loc,
                                   (float *)grad_attn, pl);
                return check_launch(g_kernel);
            }
            variant = 2;
        }
        int block = opt_bwd_block.load();
        if (block < 64 || block > 256 || (block & 63)) block = 256;  // kernels carry __launch_bounds__(256)
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
    return forward_impl<float, float>(value, shapes_dev, lstart_dev, loc, attn, N, S, M, D, L, Lq, P, out, shapes_host,
                                      (hipStream_t)stream, true);
}

int msda_forward_f64(const double *value, const int64_t *shapes_dev, const int64_t *lstart_dev, const double *loc,
                     const double *attn, int N, int S, int M, int D, int L, int Lq, int P, double *out,
                     const int64_t *shapes_host, void *stream) {
    return forward_impl<double, double>(value, shapes_dev, lstart_dev, loc, attn, N, S, M, D, L, Lq, P, out,
                                        shapes_host, (hipStream_t)stream, false);
}

int msda_forward_bf16(const uint16_t *value, const int64_t *shapes_dev, const int64_t *lstart_dev, const float *loc,
                      const float *attn, int N, int S, int M, int D, int L, int Lq, int P, uint16_t *out,
                      const int64_t *shapes_host, void *stream) {
    return forward_impl<bf16_t, float>((const bf16_t *)value, shapes_dev, lstart_dev, loc, attn, N, S, M, D, L, Lq, P,
                                       (bf16_t *)out, shapes_host, (hipStream_t)stream, false);
}

int msda_backward_f32(const float *value, const int64_t *shapes_dev, const int64_t *lstart_dev, const float *loc,
                      const float *attn, const float *grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                      float *grad_value, float *grad_loc, float *grad_attn, int zero_grad_value,
                      const int64_t *shapes_host, void *stream) {
    return backward_impl<float, float, float>(value, shapes_dev, lstart_dev, loc, attn, grad_out, N, S, M, D, L, Lq, P,
                                              grad_value, grad_loc, grad_attn, zero_grad_value, shapes_host,
                                              (hipStream_t)stream, true);
}

int msda_backward_f64(const double *value, const int64_t *shapes_dev, const int64_t *lstart_dev, const double *loc,
                      const double *attn, const double *grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                      double *grad_value, double *grad_loc, double *grad_attn, int zero_grad_value,
                      const int64_t *shapes_host, void *stream) {
    return backward_impl<double, double, double>(value, shapes_dev, lstart_dev, loc, attn, grad_out, N, S, M, D, L, Lq,
                                                 P, grad_value, grad_loc, grad_attn, zero_grad_value, shapes_host,
                                                 (hipStream_t)stream, false);
}

int msda_backward_bf16(const uint16_t *value, const int64_t *shapes_dev, const int64_t *lstart_dev, const float *loc,
                       const float *attn, const uint16_t *grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                       float *grad_value, float *grad_loc, float *grad_attn, int zero_grad_value,
                       const int64_t *shapes_host, void *stream) {
    return backward_impl<bf16_t, float, float>((const bf16_t *)value, shapes_dev, lstart_dev, loc, attn,
                                               (const bf16_t *)grad_out, N, S, M, D, L, Lq, P, grad_value, grad_loc,
                                               grad_attn, zero_grad_value, shapes_host, (hipStream_t)stream, false);
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
