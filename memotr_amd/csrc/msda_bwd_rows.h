// msda_bwd_rows.h -- backward for D = 32 calls with few queries (the decoder's cross-attention: Lq = 300 + tracks), round 3.
// Included by msda_hip.hip inside its anonymous namespace.
//
// Semantics: models/ops/src/cuda/ms_deform_im2col_cuda.cuh:301-403 (+ bilinear :87-159); fused prologue Jacobians of
// models/ops/modules/ms_deform_attn.py:104-123 (what msda_bwd_generic<.., FUSED> computes, same formulas).
//
// The generic kernel spends one 64-thread block (half of it idle at D = 32) per (query, head) row and walks its
// points one by one: value load -> atomic -> three wave reductions, 16 dependent round trips per row.  Measured 41-55 us
// for 2,560 rows: L2 float-atomic bound at the slow "8 rows x 8 lanes, 16-byte stride" pattern (83 G atomics/s,
// profiles/r01_ubench_atomics_l1.txt).  Here:
//   * 32 lanes own a row (one channel each): a corner's 32 atomics cover one 128-byte row of grad_value -- the
//     "2 x 32 contiguous" pattern of the L2 atomic units (322 G/s);
//   * the row's points are staged once (one lane per point: location, corner offsets with out-of-range markers for
//     corners that do not exist, so loads return 0 and atomics are dropped without branches);
//   * four points (16 value loads per lane) are in flight together; the three per-point sums reduce over the 32
//     lanes with DPP inside 16 lanes and one cross-row exchange;
//   * the fused epilogue (softmax and location Jacobians, reference-point partials) is the generic kernel's.
#pragma once

constexpr int kRowsMaxLP = 32;
constexpr unsigned kRowsOobElem = 0x30000000u;      // x 4 (fp32 / grad_value bytes) and x 2 (bf16 bytes) stay out of range

template <typename TV>
__device__ __forceinline__ float rows_load(__amdgpu_buffer_rsrc_t r, unsigned elem);
template <>
__device__ __forceinline__ float rows_load<float>(__amdgpu_buffer_rsrc_t r, unsigned elem) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)(elem * 4u), 0, 0));
}
template <>
__device__ __forceinline__ float rows_load<bf16_t>(__amdgpu_buffer_rsrc_t r, unsigned elem) {
    const unsigned short u = __builtin_amdgcn_raw_buffer_load_b16(r, (int)(elem * 2u), 0, 0);
    return __uint_as_float(((unsigned)u) << 16);
}

__device__ __forceinline__ float half32_sum(float x) {       // over the 32 lanes that own a row
    x += MSDA_DPP(x, 0xB1);
    x += MSDA_DPP(x, 0x4E);
    x += MSDA_DPP(x, 0x141);
    x += MSDA_DPP(x, 0x140);
    x += __shfl_xor(x, 16, kWave);
    return x;
}
__device__ __forceinline__ float half32_max(float x) {
    x = fmaxf(x, MSDA_DPP(x, 0xB1));
    x = fmaxf(x, MSDA_DPP(x, 0x4E));
    x = fmaxf(x, MSDA_DPP(x, 0x141));
    x = fmaxf(x, MSDA_DPP(x, 0x140));
    x = fmaxf(x, __shfl_xor(x, 16, kWave));
    return x;
}

// ATOMICS = false (round 6): everything but grad_value -- the sorted backward (msda_bwd_sorted.h) builds grad_value by a
// sort + gather and takes grad_loc / grad_attn / the fused Jacobians from here; `zero` / `zero_n`: words that launch
// clears on the side (the sort's bucket totals, which its counting kernel then adds to).
template <typename TV, bool FUSED, bool ATOMICS = true>
__global__ __launch_bounds__(256) void msda_bwd_d32_rows(
    const TV *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lstart,
    const PointSrc fs, const TV *__restrict__ grad_out, int N, int S, int M, int L, int Lq, int P,
    float *__restrict__ grad_value, float *__restrict__ grad_loc, float *__restrict__ grad_attn,
    float *__restrict__ grad_proj, float *__restrict__ grad_ref_part, unsigned value_bytes, unsigned gv_bytes,
    unsigned *__restrict__ zero = nullptr, unsigned zero_n = 0u, unsigned pix_elems = 0u) {
    // pix_elems: elements between two pixels of `value` AND of `grad_value` (0: M * D, contiguous tensors); larger when
    // both are slices of a wider projection and of its gradient (msda_next_value_pixel_stride)
    constexpr unsigned D = 32;
    if (!ATOMICS && zero != nullptr)
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < zero_n; i += gridDim.x * blockDim.x) zero[i] = 0u;
    __shared__ int s_H[kMaxLevels], s_W[kMaxLevels], s_start[kMaxLevels];
    __shared__ u32x4 s_rec[8][kRowsMaxLP][2];      // per row slot and point: {4 corner elements} {lh, lw, a, gate}
    __shared__ float s_res[8][3 * kRowsMaxLP];     // per row slot: d/dx, d/dy (2t, 2t+1), d/dattn (2 LP + t)
    if (threadIdx.x < L) {
        s_H[threadIdx.x] = (int)shapes[2 * threadIdx.x];
        s_W[threadIdx.x] = (int)shapes[2 * threadIdx.x + 1];
        s_start[threadIdx.x] = (int)lstart[threadIdx.x];
    }
    __syncthreads();
    const int LP = L * P;
    const int slot = threadIdx.x >> 5, c = threadIdx.x & 31;
    const unsigned n_rows = (unsigned)N * (unsigned)Lq * (unsigned)M;
    const unsigned row_elems = pix_elems ? pix_elems : (unsigned)M * D;
    const __amdgpu_buffer_rsrc_t vr = make_rsrc(value, value_bytes);
    const __amdgpu_buffer_rsrc_t gr = make_rsrc(grad_value, gv_bytes);
    u32x4 (*rec)[2] = s_rec[slot];
    float *res = s_res[slot];
    const unsigned slots = blockDim.x >> 5;             // rows per workgroup pass (2 for 64-thread workgroups)
    const unsigned n_groups = (n_rows + slots - 1u) / slots;
    for (unsigned grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const unsigned pm0 = grp * slots + (unsigned)slot;
        const bool row_ok = pm0 < n_rows;
        const unsigned pm = row_ok ? pm0 : n_rows - 1;
        const unsigned qrow = pm / (unsigned)M;
        const int m = (int)(pm - qrow * (unsigned)M);
        const unsigned b = qrow / (unsigned)Lq;
        // ---- stage this lane's point (lane t < LP) ----
        float mx = 0.f, rsum = 1.f, e_t = 0.f;
        if (FUSED) {
            const float lg = c < LP ? fused_logits(fs, qrow, m, LP)[c] : -INFINITY;
            mx = half32_max(lg);
            e_t = sm_exp(lg, mx);                   // (msda_common.h: the one softmax arithmetic; half32_sum IS its tree)
            rsum = sm_rcp(half32_sum(e_t));
        }
        if (c < LP) {
            const int t = c, l = t / P;
            const int H = s_H[l], W = s_W[l];
            const f32x2 xy = point_location<FUSED>(fs, pm, qrow, m, L, P, t, l, H, W);
            const float a = FUSED ? e_t * rsum : fs.attn[pm * (unsigned)LP + (unsigned)t];
            const Sample<float> s = sample_setup<float>(xy.x, xy.y, H, W);
            const bool live = s.gate && row_ok;
            const int h0 = s.h_low, w0 = s.w_low, h1 = h0 + 1, w1 = w0 + 1;
            bool ok1 = live && h0 >= 0 && w0 >= 0, ok2 = live && h0 >= 0 && w1 <= W - 1;
            bool ok3 = live && h1 <= H - 1 && w0 >= 0, ok4 = live && h1 <= H - 1 && w1 <= W - 1;
            if (FUSED && fs.mask != nullptr) {   // padded pixels: value reads as 0 and receives no gradient
                const unsigned char *mk = fs.mask + ((size_t)b * S + s_start[l]);
                const int p00 = h0 * W + w0;
                ok1 = ok1 && !mk[ok1 ? p00 : 0];
                ok2 = ok2 && !mk[ok2 ? p00 + 1 : 0];
                ok3 = ok3 && !mk[ok3 ? p00 + W : 0];
                ok4 = ok4 && !mk[ok4 ? p00 + W + 1 : 0];
            }
            const unsigned e1 = (b * (unsigned)S + (unsigned)(s_start[l] + h0 * W + w0)) * row_elems + (unsigned)m * D;
            u32x4 off, w;
            off.x = ok1 ? e1 : kRowsOobElem;
            off.y = ok2 ? e1 + row_elems : kRowsOobElem;
            off.z = ok3 ? e1 + (unsigned)W * row_elems : kRowsOobElem;
            off.w = ok4 ? e1 + (unsigned)W * row_elems + row_elems : kRowsOobElem;
            w.x = __float_as_uint(live ? s.lh : 0.f);
            w.y = __float_as_uint(live ? s.lw : 0.f);
            w.z = __float_as_uint(live ? a : 0.f);
            w.w = live ? 1u : 0u;
            rec[t][0] = off;
            rec[t][1] = w;
        }
        const float g = row_ok ? to_compute(grad_out[pm * D + (unsigned)c]) : 0.f;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // ---- four points at a time: 16 value loads per lane in flight, then atomics and the three sums ----
        for (int t0 = 0; t0 < LP; t0 += 4) {
            u32x4 off[4], w[4];
            float v[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = t0 + i < LP ? t0 + i : LP - 1;
                off[i] = rec[t][0];
                w[i] = rec[t][1];
                if (t0 + i >= LP) {       // past the end: inert
                    off[i] = u32x4{kRowsOobElem, kRowsOobElem, kRowsOobElem, kRowsOobElem};
                    w[i] = u32x4{0u, 0u, 0u, 0u};
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i][0] = rows_load<TV>(vr, off[i].x + (unsigned)c);
                v[i][1] = rows_load<TV>(vr, off[i].y + (unsigned)c);
                v[i][2] = rows_load<TV>(vr, off[i].z + (unsigned)c);
                v[i][3] = rows_load<TV>(vr, off[i].w + (unsigned)c);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float lh = __uint_as_float(w[i].x), lw = __uint_as_float(w[i].y), a = __uint_as_float(w[i].z);
                const bool gate = w[i].w != 0u;
                const float hh = 1.f - lh, hw = 1.f - lw;
                const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
                const float tga = g * a;
                if (ATOMICS) {
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(w1 * tga, gr, (int)((off[i].x + (unsigned)c) * 4u), 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(w2 * tga, gr, (int)((off[i].y + (unsigned)c) * 4u), 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(w3 * tga, gr, (int)((off[i].z + (unsigned)c) * 4u), 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(w4 * tga, gr, (int)((off[i].w + (unsigned)c) * 4u), 0, 0);
                }
                const float gw = hh * (v[i][1] - v[i][0]) + lh * (v[i][3] - v[i][2]);
                const float gh = hw * (v[i][2] - v[i][0]) + lw * (v[i][3] - v[i][1]);
                // a gated-off point contributes exactly nothing, also for non-finite gradients
                const float pa = gate ? g * (w1 * v[i][0] + w2 * v[i][1] + w3 * v[i][2] + w4 * v[i][3]) : 0.f;
                const float px = gate ? gw * tga : 0.f, py = gate ? gh * tga : 0.f;
                const float sa = half32_sum(pa), sx = half32_sum(px), sy = half32_sum(py);
                if (c == 0 && t0 + i < LP) {
                    res[2 * (t0 + i)] = sx;
                    res[2 * (t0 + i) + 1] = sy;
                    res[2 * LP + t0 + i] = sa;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // ---- per-point results: scale by the level size; fused: the prologue's Jacobians ----
        if (row_ok && c < LP) {
            const int t = c, l = t / P;
            const float gx = res[2 * t] * (float)s_W[l], gy = res[2 * t + 1] * (float)s_H[l];
            if (!FUSED) {
                *reinterpret_cast<f32x2 *>(grad_loc + ((size_t)pm * LP + t) * 2) = f32x2{gx, gy};
                grad_attn[(size_t)pm * LP + t] = res[2 * LP + t];
            } else {
                const float *lg = fused_logits(fs, qrow, m, LP);
                float *gp = grad_proj + (size_t)qrow * fs.proj_stride;
                float dot = 0.f;
                for (int j = 0; j < LP; ++j) dot += (sm_exp(lg[j], mx) * rsum) * res[2 * LP + j];
                const float a_t = sm_exp(lg[t], mx) * rsum;
                gp[fs.n_off + m * LP + t] = a_t * (res[2 * LP + t] - dot);
                if (fs.ref_dim == 2) {
                    gp[(m * LP + t) * 2] = gx / (float)s_W[l];
                    gp[(m * LP + t) * 2 + 1] = gy / (float)s_H[l];
                } else {
                    const float *r = fs.ref + ((size_t)qrow * L + l) * 4;
                    gp[(m * LP + t) * 2] = gx * (r[2] * (0.5f / (float)P));
                    gp[(m * LP + t) * 2 + 1] = gy * (r[3] * (0.5f / (float)P));
                }
            }
        }
        if (FUSED && grad_ref_part != nullptr && row_ok && c < L * fs.ref_dim) {
            const int l = c / fs.ref_dim, comp = c - l * fs.ref_dim;
            const float *off = fs.proj + (size_t)qrow * fs.proj_stride + ((size_t)m * LP + l * P) * 2;
            const float scale = (comp & 1) ? (float)s_H[l] : (float)s_W[l];
            float acc = 0.f;
            for (int p = 0; p < P; ++p) {
                const float gl = res[2 * (l * P + p) + (comp & 1)] * scale;
                acc += comp < 2 ? gl : gl * off[2 * p + (comp & 1)] * (0.5f / (float)P);
            }
            grad_ref_part[((size_t)pm * L + l) * fs.ref_dim + comp] = acc;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}
