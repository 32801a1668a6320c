// msda_fused_side.h -- side kernels of the fused backward: softmax / location Jacobians applied in place.
// Included by msda_hip.hip inside its anonymous namespace.
#pragma once

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
        for (int t = sub; t < LP; t += 8) dot += (sm_exp(lg[t], mx) * rsum) * ga[t];
        dot = row_sum<8>(dot);
        if (ok)
            for (int t = sub; t < LP; t += 8) ga[t] = (sm_exp(lg[t], mx) * rsum) * (ga[t] - dot);
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
// Jacobian run in registers -- no cross-lane traffic, a sixteenth of the threads, one index division per row.  The
// softmax is msda_common.h's (sm_exp / sm_rcp, adjacent-pair tree); the Jacobian's dot product associates like the 16-lane
// butterflies above ((t, t + 8) first, then 4, then (0 + 2) + (1 + 3)), so the bits are theirs.
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
        float e[16], q[4];
#pragma unroll
        for (int t = 0; t < 16; ++t) e[t] = sm_exp(lg[t], mx);
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = (e[4 * k] + e[4 * k + 1]) + (e[4 * k + 2] + e[4 * k + 3]);
        const float rsum = sm_rcp((q[0] + q[1]) + (q[2] + q[3]));       // (the adjacent-pair tree of msda_common.h)
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
