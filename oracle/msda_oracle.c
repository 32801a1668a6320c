/*
 * oracle/msda_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference's multi-scale deformable attention
 * operator (sampling-point gather + bilinear interpolation + attention-weight
 * reduction), forward and backward.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product path
 * (memotr_amd/) never does.
 *
 * What each function follows in the reference (paths under /root/reference):
 *   sample_setup_*        models/ops/src/cuda/ms_deform_im2col_cuda.cuh:285-288
 *                         (h_im/w_im and the (-1,H)x(-1,W) gate) and :38-46
 *                         (floor, fractional parts)
 *   msda_oracle_forward_* .cuh:237-299 (per-output loop over L x P) with the
 *                         4-corner zero-padded bilinear read of .cuh:33-84
 *   msda_oracle_backward_*.cuh:301-403 (per-(l,p) reduction over channels of
 *                         grad_sampling_loc / grad_attn_weight) with the
 *                         corner scatter of .cuh:87-159
 *   msda_oracle_indices_* the integer part only (floor results + gate), used
 *                         to check the HIP kernels' index arithmetic bit-exactly
 *
 * Layouts (all contiguous, as asserted by ms_deform_attn_cuda.cu:28-38):
 *   value (N,S,M,D)  shapes (L,2)=(H,W) int64  level_start (L,) int64
 *   loc (N,Lq,M,L,P,2) in (x,y) order   attn (N,Lq,M,L,P)   out (N,Lq,M*D)
 *
 * Pinned against tests/golden/ (vectors produced by importing the reference's
 * own ms_deform_attn_core_pytorch + autograd, models/ops/functions/
 * ms_deform_attn_func.py:44-64; generator: tests/golden/gen_golden.py).
 *
 * Build: see oracle/Makefile (-ffp-contract=off so that the float variant
 * keeps the reference's rounding points: loc*size is rounded before -0.5).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define DEFINE_ORACLE(T, SUF)                                                          \
                                                                                       \
    typedef struct {                                                                   \
        int gate;      /* point contributes at all (.cuh:288) */                       \
        int h_low, w_low;                                                              \
        T lh, lw;                                                                      \
    } sample_##SUF;                                                                    \
                                                                                       \
    static sample_##SUF sample_setup_##SUF(T loc_w, T loc_h, int H, int W) {           \
        sample_##SUF s;                                                                \
        /* .cuh:285-286: product rounded in T, then minus one half */                  \
        const T h_im = (T)(loc_h * (T)H) - (T)0.5;                                     \
        const T w_im = (T)(loc_w * (T)W) - (T)0.5;                                     \
        s.gate = (h_im > (T)-1 && w_im > (T)-1 && h_im < (T)H && w_im < (T)W);         \
        s.h_low = (int)floor((double)h_im);                                            \
        s.w_low = (int)floor((double)w_im);                                            \
        s.lh = h_im - (T)s.h_low;                                                      \
        s.lw = w_im - (T)s.w_low;                                                      \
        return s;                                                                      \
    }                                                                                  \
                                                                                       \
    void msda_oracle_indices_##SUF(const int64_t *shapes, const T *loc, int N, int M,  \
                                   int L, int Lq, int P, int32_t *h_low,               \
                                   int32_t *w_low, uint8_t *gate) {                    \
        const long n_pairs = (long)N * Lq * M;                                         \
        for (long pm = 0; pm < n_pairs; ++pm)                                          \
            for (int l = 0; l < L; ++l) {                                              \
                const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];          \
                for (int p = 0; p < P; ++p) {                                          \
                    const long t = (pm * L + l) * P + p;                               \
                    sample_##SUF s = sample_setup_##SUF(loc[2 * t], loc[2 * t + 1], H, W); \
                    h_low[t] = s.h_low;                                                \
                    w_low[t] = s.w_low;                                                \
                    gate[t] = (uint8_t)s.gate;                                         \
                }                                                                      \
            }                                                                          \
    }                                                                                  \
                                                                                       \
    void msda_oracle_forward_##SUF(const T *value, const int64_t *shapes,              \
                                   const int64_t *level_start, const T *loc,           \
                                   const T *attn, int N, int S, int M, int D, int L,   \
                                   int Lq, int P, T *out) {                            \
        const long row = (long)M * D; /* stride between neighbouring pixels */         \
        for (int b = 0; b < N; ++b)                                                    \
            for (int q = 0; q < Lq; ++q)                                               \
                for (int m = 0; m < M; ++m) {                                          \
                    const long pm = ((long)b * Lq + q) * M + m;                        \
                    T *o = out + pm * D;                                               \
                    for (int c = 0; c < D; ++c) o[c] = (T)0;                           \
                    for (int l = 0; l < L; ++l) {                                      \
                        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];  \
                        const T *v = value + ((long)b * S + level_start[l]) * row + (long)m * D; \
                        for (int p = 0; p < P; ++p) {                                  \
                            const long t = (pm * L + l) * P + p;                       \
                            sample_##SUF s = sample_setup_##SUF(loc[2 * t], loc[2 * t + 1], H, W); \
                            if (!s.gate) continue;                                     \
                            const T a = attn[t];                                       \
                            const T hh = (T)1 - s.lh, hw = (T)1 - s.lw;                \
                            const T w1 = hh * hw, w2 = hh * s.lw, w3 = s.lh * hw, w4 = s.lh * s.lw; \
                            const int h0 = s.h_low, w0 = s.w_low, h1 = h0 + 1, w1i = w0 + 1; \
                            const int ok1 = (h0 >= 0 && w0 >= 0);                      \
                            const int ok2 = (h0 >= 0 && w1i <= W - 1);                 \
                            const int ok3 = (h1 <= H - 1 && w0 >= 0);                  \
                            const int ok4 = (h1 <= H - 1 && w1i <= W - 1);             \
                            for (int c = 0; c < D; ++c) {                              \
                                const T v1 = ok1 ? v[((long)h0 * W + w0) * row + c] : (T)0;  \
                                const T v2 = ok2 ? v[((long)h0 * W + w1i) * row + c] : (T)0; \
                                const T v3 = ok3 ? v[((long)h1 * W + w0) * row + c] : (T)0;  \
                                const T v4 = ok4 ? v[((long)h1 * W + w1i) * row + c] : (T)0; \
                                const T val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;   \
                                o[c] += val * a;                                       \
                            }                                                          \
                        }                                                              \
                    }                                                                  \
                }                                                                      \
    }                                                                                  \
                                                                                       \
    /* grad_value must be zero-filled by the caller (the reference allocates it     */ \
    /* with zeros_like, ms_deform_attn_cuda.cu:121); grad_loc / grad_attn are       */ \
    /* fully overwritten.                                                           */ \
    void msda_oracle_backward_##SUF(const T *value, const int64_t *shapes,             \
                                    const int64_t *level_start, const T *loc,          \
                                    const T *attn, const T *grad_out, int N, int S,    \
                                    int M, int D, int L, int Lq, int P, T *grad_value, \
                                    T *grad_loc, T *grad_attn) {                       \
        const long row = (long)M * D;                                                  \
        for (int b = 0; b < N; ++b)                                                    \
            for (int q = 0; q < Lq; ++q)                                               \
                for (int m = 0; m < M; ++m) {                                          \
                    const long pm = ((long)b * Lq + q) * M + m;                        \
                    const T *g = grad_out + pm * D;                                    \
                    for (int l = 0; l < L; ++l) {                                      \
                        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];  \
                        const long base = ((long)b * S + level_start[l]) * row + (long)m * D; \
                        const T *v = value + base;                                     \
                        T *gv = grad_value + base;                                     \
                        for (int p = 0; p < P; ++p) {                                  \
                            const long t = (pm * L + l) * P + p;                       \
                            sample_##SUF s = sample_setup_##SUF(loc[2 * t], loc[2 * t + 1], H, W); \
                            T acc_w = (T)0, acc_h = (T)0, acc_a = (T)0;                \
                            if (s.gate) {                                              \
                                const T a = attn[t];                                   \
                                const T hh = (T)1 - s.lh, hw = (T)1 - s.lw;            \
                                const T w1 = hh * hw, w2 = hh * s.lw, w3 = s.lh * hw, w4 = s.lh * s.lw; \
                                const int h0 = s.h_low, w0 = s.w_low, h1 = h0 + 1, w1i = w0 + 1; \
                                const int ok1 = (h0 >= 0 && w0 >= 0);                  \
                                const int ok2 = (h0 >= 0 && w1i <= W - 1);             \
                                const int ok3 = (h1 <= H - 1 && w0 >= 0);              \
                                const int ok4 = (h1 <= H - 1 && w1i <= W - 1);         \
                                for (int c = 0; c < D; ++c) {                          \
                                    const T top = g[c];                                \
                                    const T tga = top * a;                             \
                                    T gh = (T)0, gw = (T)0;                            \
                                    T v1 = (T)0, v2 = (T)0, v3 = (T)0, v4 = (T)0;      \
                                    if (ok1) {                                         \
                                        const long i = ((long)h0 * W + w0) * row + c;  \
                                        v1 = v[i]; gh -= hw * v1; gw -= hh * v1;       \
                                        gv[i] += w1 * tga;                             \
                                    }                                                  \
                                    if (ok2) {                                         \
                                        const long i = ((long)h0 * W + w1i) * row + c; \
                                        v2 = v[i]; gh -= s.lw * v2; gw += hh * v2;     \
                                        gv[i] += w2 * tga;                             \
                                    }                                                  \
                                    if (ok3) {                                         \
                                        const long i = ((long)h1 * W + w0) * row + c;  \
                                        v3 = v[i]; gh += hw * v3; gw -= s.lh * v3;     \
                                        gv[i] += w3 * tga;                             \
                                    }                                                  \
                                    if (ok4) {                                         \
                                        const long i = ((long)h1 * W + w1i) * row + c; \
                                        v4 = v[i]; gh += s.lw * v4; gw += s.lh * v4;   \
                                        gv[i] += w4 * tga;                             \
                                    }                                                  \
                                    const T val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4; \
                                    acc_a += top * val;                                \
                                    acc_w += (T)W * gw * tga;                          \
                                    acc_h += (T)H * gh * tga;                          \
                                }                                                      \
                            }                                                          \
                            grad_loc[2 * t] = acc_w;                                   \
                            grad_loc[2 * t + 1] = acc_h;                               \
                            grad_attn[t] = acc_a;                                      \
                        }                                                              \
                    }                                                                  \
                }                                                                      \
    }

DEFINE_ORACLE(float, f32)
DEFINE_ORACLE(double, f64)

/* The OTHER reading of .cuh:285-286.  `loc_h * spatial_h - 0.5` is one multiply and one subtract in the source.  Whether
 * the reference BINARY fuses them is UNDECIDED without nvcc (absent from this image): setup.py
 * (models/ops/setup.py:41-46) passes no -fmad=false, which would let nvcc contract a same-type multiply + add; but the
 * literal 0.5 is a double, so for scalar_t = float the expression is fptrunc(fpext(loc * size) - 0.5) -- a float
 * multiply feeding a double subtract -- which a compiler may or may not narrow back to a float fma.  The oracle above
 * and the HIP kernels (`fp contract(off)`) follow the uncontracted source.  This probe computes the contracted indices
 * so that tests/test_oracle_golden.py can COUNT the points that flip at the BASELINE shapes: zero of 2.86 M corner
 * indices on either benchmark distribution; 17,063, each by one pixel with a bilinear weight <= 2^-23 on the far side,
 * for locations placed exactly on pixel centres. */
void msda_oracle_indices_fma_f32(const int64_t *shapes, const float *loc, int N, int M, int L, int Lq, int P,
                                 int32_t *h_low, int32_t *w_low, uint8_t *gate) {
    const long n_pairs = (long)N * Lq * M;
    for (long pm = 0; pm < n_pairs; ++pm)
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            for (int p = 0; p < P; ++p) {
                const long t = (pm * L + l) * P + p;
                const float h_im = fmaf(loc[2 * t + 1], (float)H, -0.5f);
                const float w_im = fmaf(loc[2 * t], (float)W, -0.5f);
                h_low[t] = (int32_t)floor((double)h_im);
                w_low[t] = (int32_t)floor((double)w_im);
                gate[t] = (uint8_t)(h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W);
            }
        }
}

