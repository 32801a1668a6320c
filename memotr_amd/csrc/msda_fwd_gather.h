// msda_fwd_gather.h -- forward for D = 32 through the vector L1 (decoder queries, bf16 rows, pyramids whose points are
// far from their queries).  Included by msda_hip.hip inside its anonymous namespace.
#pragma once

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
                                                  unsigned row_base, unsigned pix_stride, const int *s_H,
                                                  const int *s_W, const int *s_start) {
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
            // (the one softmax arithmetic of the fused kernels, msda_common.h: chunk butterflies, then the chunks in order)
            mx = row_max<LANES>(fmaxf(l0, l1));
            e0 = sm_exp(l0, mx);
            e1 = sm_exp(l1, mx);
            rsum = sm_rcp(row_sum<LANES>(e0) + row_sum<LANES>(e1));
        } else {
            row_softmax_stats<LANES>(lg, LP, sub, mx, rsum);
        }
    }
    const float rcp_p = 1.f / (float)P;
    for (int t = sub; t < LP; t += LANES) {
        const int l = (int)(((float)t + 0.5f) * rcp_p);      // == t / P (the product stays 0.5/P away from integers)
        const int H = s_H[l], W = s_W[l];
        const f32x2 xy = point_location<FUSED>(src, pmc, qrow, m, L, P, t, l, H, W);
        const float a_in = FUSED ? (two ? (t == sub ? e0 : e1) : sm_exp(lg[t], mx)) * rsum
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
    int head_major, unsigned pix_bytes) {
    // pix_bytes: bytes between two pixels of `value` -- M * D * sizeof(TV) for a contiguous (N, S, M, D) tensor, more when
    // the rows are a slice of a wider projection (msda_next_value_pixel_stride, round 6: the decoder layers' six value
    // projections as ONE GEMM whose output each layer reads in place)
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
        const unsigned row_base = (unsigned)b * (unsigned)S * pix_bytes + (unsigned)m * (D * (unsigned)sizeof(TV));
        stage_records_fwd<TV, FUSED>(rec, src, pmc, qrow, m, row_ok, sub, L, P, M, S, b, row_base, pix_bytes, s_H, s_W, s_start);
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
