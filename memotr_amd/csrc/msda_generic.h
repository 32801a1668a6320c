// msda_generic.h -- the kernels for any D / dtype (one thread per output scalar), the parity hooks and the fused
// prologue's stand-alone kernels.  Included by msda_hip.hip inside its anonymous namespace.
#pragma once

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
                attn_out[pm * LP + t] = sm_exp(lg[t], mx) * rsum;
            }
        }
    }
}

// The same for L*P <= 16 with one lane per point (16 lanes per row): every lane reads and writes consecutive
// addresses.  The softmax is the one of msda_common.h (sm_exp / sm_rcp, adjacent-pair tree): the same bits as the 8-lane
// form above and as every forward and backward kernel.
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
        const float e = sm_exp(lg, mx);
        float sum = e + __shfl_xor(e, 1, 16);       // (the adjacent-pair tree of msda_common.h)
        sum += __shfl_xor(sum, 2, 16);
        sum += __shfl_xor(sum, 4, 16);
        sum += __shfl_xor(sum, 8, 16);
        const float rsum = sm_rcp(sum);
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

// Zero `n` 32-bit words at `p` (16-byte aligned: a tensor) and `extra_n` words at `extra`: grad_value on request, and what
// the chosen backward wants cleared next to it.
__global__ __launch_bounds__(256) void msda_zero_words_kernel(unsigned *__restrict__ p, size_t n, unsigned *__restrict__ extra,
                                                              unsigned extra_n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((reinterpret_cast<uintptr_t>(p) & 15u) == 0u) {
        u32x4 *const p4 = reinterpret_cast<u32x4 *>(p);
        for (size_t i = i0; i < n / 4; i += stride) p4[i] = u32x4{0u, 0u, 0u, 0u};
        for (size_t i = (n & ~(size_t)3) + i0; i < n; i += stride) p[i] = 0u;
    } else {
        for (size_t i = i0; i < n; i += stride) p[i] = 0u;
    }
    for (size_t i = i0; i < extra_n; i += stride) extra[i] = 0u;
}
