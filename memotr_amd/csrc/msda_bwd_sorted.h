// msda_bwd_sorted.h -- grad_value of the backward WITHOUT fabric atomics, for sampling points far from their queries
// (selector level 2: most corners leave even the large window of msda_bwd_d32_bins), round 6.
// Included by msda_hip.hip inside its anonymous namespace.
//
// Reference semantics: ms_deformable_col2im_gpu_kernel_* + ms_deform_attn_col2im_bilinear
// (models/ops/src/cuda/ms_deform_im2col_cuda.cuh:87-159, 301-403): every (query, head, level, point, corner) adds
// w_corner * attn * grad_out_row to one 128-byte row of grad_value.  The reference does that with 32 atomicAdds per
// corner and costs the same wherever the points land; so did round 3's msda_bwd_d32_rows (whole-row buffer atomics),
// which for the encoder call is 11.4 M corner rows = 366 M lane-atomics = 1.09 ms of L2 atomic units whatever else the
// kernel does (profiles/r05_msda_bench_rocprof_summary.md: fetch 768 MB + write 1,454 MB = 16 x the algorithmic bytes).
//
// Here the scatter becomes a SORT followed by a GATHER, like msda_bwd_d32_bins does inside one window -- but over the
// whole tensor, through a workspace, so that nothing depends on where the points land:
//
//   count   one lane per (query, point) of one head: sampling arithmetic, the four corners' destination pixels; the
//           destination rows of a (batch, head) are cut into BUCKETS of 64 consecutive pixels; a workgroup (one head, a
//           chunk of queries) histograms its corners per bucket in LDS, stores the histogram (its own row of a count
//           matrix) and adds it to the bucket totals (one fire-and-forget integer atomic per non-empty bucket);
//   scan    per (batch, head): bucket totals -> bucket starts (exclusive prefix; every (batch, head) owns a fixed
//           segment of the record array, so no prefix crosses workgroups), and the gather's WORK LIST: a bucket with
//           more records than one workgroup sorts in LDS is cut into equal slices;
//   emit    the count kernel again, now writing: a workgroup reserves its share of every bucket with ONE returning
//           atomic per bucket (its own count from the matrix), then every corner takes an LDS ticket and stores an
//           8-byte record {query << 6 | pixel inside the bucket, w_corner * attn} -- 91 MB for the encoder call;
//   gather  a workgroup per work item: counting sort of the slice's records by destination pixel in LDS (64 counters,
//           integer tickets), then four lanes x eight channels per destination row walk the row's list:
//           acc += w * grad_out[query, head, :] (16-byte loads, one head per XCD so that its 2.9 MB grad_out slab stays
//           in that XCD's L2), fp32 accumulation in registers, and ONE plain 128-byte store per row -- atomics only for
//           the rows of buckets that were sliced (coarse pyramid levels, where thousands of points share a pixel).
//
// grad_loc / grad_attn (and the fused prologue's Jacobians) are the dot products of msda_bwd_d32_rows, instantiated
// without its atomics.  Non-finite gradients propagate through fp32 arithmetic like the reference's atomicAdd.
#pragma once

constexpr int kSortBPLog = 6, kSortBP = 1 << kSortBPLog;      // destination pixels per bucket
constexpr int kSortThreads = 256;
constexpr int kSortSlice = 4608;         // records one gathering workgroup sorts (36 KB of LDS: four workgroups per CU)
constexpr int kSortRecPerThread = kSortSlice / kSortThreads;
constexpr int kSortMaxBuckets = 12288;   // per (batch, head): the histogram lives in LDS (48 KB)

struct SortPlan {
    int N, S, M, L, Lq, P;
    int qc;                  // queries per counting / emitting workgroup
    int nchunk;              // ... workgroups per (batch, head): ceil(Lq / qc)
    int nbk;                 // buckets per (batch, head): ceil(S / 64)
    int magic_lp;            // (i * magic_lp) >> 16 == i / (L * P) for i < qc * L * P
    unsigned cap;            // records per (batch, head) segment: Lq * L * P * 4
    int max_items;           // work items per (batch, head), upper bound
    unsigned *cnt;           // [N * M][nchunk][nbk] corner counts per counting workgroup and bucket
    unsigned *cursor;        // [N * M][nbk] bucket totals (count), then the next free record of the bucket (scan, emit)
    unsigned *nwork;         // [N * M]
    u32x4 *work;             // [N * M][max_items] {bucket, first record, records, sliced}
    u32x2 *rec;              // [N * M][cap]
    size_t bytes;            // of the whole scratch block
};

// Layout of the scratch block; false when the call does not fit the kernels' 32-bit arithmetic.
inline bool make_sort_plan(SortPlan &sp, int N, int S, int M, int L, int Lq, int P, size_t elem_bytes, void *workspace) {
    memset(&sp, 0, sizeof(sp));
    const long LP = (long)L * P;
    if (LP < 1 || LP > kRowsMaxLP || N < 1 || Lq < 1 || Lq >= (1 << 26)) return false;
    const long nbk = ((long)S + kSortBP - 1) >> kSortBPLog;
    if (nbk > kSortMaxBuckets) return false;
    const long cap = (long)Lq * LP * 4, total = cap * N * M;
    if (total >= (1L << 31) || (long)N * Lq * M * 32 * (long)elem_bytes >= 0x7fffff00L) return false;
    // enough counting workgroups to fill the chip, few enough that the count matrix stays small
    int qc = 256;
    while (qc > 32 && (long)N * M * ((Lq + qc - 1) / qc) < 1024) qc >>= 1;
    const int items = qc * (int)LP;
    const int magic = 65536 / (int)LP + 1;
    for (int i = 0; i < items; ++i)
        if (((i * magic) >> 16) != i / (int)LP) return false;
    sp.N = N; sp.S = S; sp.M = M; sp.L = L; sp.Lq = Lq; sp.P = P;
    sp.qc = qc; sp.nchunk = (Lq + qc - 1) / qc; sp.nbk = (int)nbk; sp.magic_lp = magic; sp.cap = (unsigned)cap;
    sp.max_items = (int)(cap / kSortSlice + nbk);
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t nbm = (size_t)N * M;
    size_t o = 0;
    unsigned char *base = reinterpret_cast<unsigned char *>(workspace);
    sp.cursor = reinterpret_cast<unsigned *>(base + o); o += up(nbm * nbk * 4);
    sp.nwork = reinterpret_cast<unsigned *>(base + o); o += up(nbm * 4);
    sp.cnt = reinterpret_cast<unsigned *>(base + o); o += up(nbm * sp.nchunk * nbk * 4);
    sp.work = reinterpret_cast<u32x4 *>(base + o); o += up(nbm * sp.max_items * 16);
    sp.rec = reinterpret_cast<u32x2 *>(base + o); o += up(nbm * cap * 8);
    sp.bytes = o;
    return true;
}

// ---- count (EMIT = false) / emit (EMIT = true): one lane per (query, point) of one head ----
template <bool EMIT>
__global__ __launch_bounds__(kSortThreads) void msda_bwd_sort_points(const int64_t *__restrict__ shapes,
                                                                      const int64_t *__restrict__ lstart,
                                                                      const float *__restrict__ loc,
                                                                      const float *__restrict__ attn,
                                                                      const unsigned char *__restrict__ mask,
                                                                      const SortPlan sp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    __shared__ int s_H[kMaxLevels], s_W[kMaxLevels], s_start[kMaxLevels];
    unsigned *const hist = reinterpret_cast<unsigned *>(s_dyn);         // count: histogram; emit: next free record
    const int tid = threadIdx.x;
    // block -> (head, batch, chunk): head-major, one head per XCD like the gather (the records of a head are written
    // and read through the same L2)
    const int per = sp.N * sp.nchunk;                   // workgroups per head
    const int n_blocks = per * sp.M;
    const int chunk8 = (int)(gridDim.x >> 3);
    const int sw = (int)(blockIdx.x & 7) * chunk8 + (int)(blockIdx.x >> 3);
    if (sw >= n_blocks) return;
    const int m = sw / per, rem = sw - m * per;
    const int b = rem / sp.nchunk, ck = rem - b * sp.nchunk;
    const int bm = b * sp.M + m;
    const int L = sp.L, P = sp.P, LP = L * P, nbk = sp.nbk;
    if (tid < L) {
        s_H[tid] = (int)shapes[2 * tid];
        s_W[tid] = (int)shapes[2 * tid + 1];
        s_start[tid] = (int)lstart[tid];
    }
    unsigned *const my_cnt = sp.cnt + ((size_t)bm * sp.nchunk + ck) * nbk;
    unsigned *const cursor = sp.cursor + (size_t)bm * nbk;
    if (EMIT) {
        // this workgroup's share of every bucket: one returning atomic per non-empty bucket
        for (int i = tid; i < nbk; i += kSortThreads) {
            const unsigned c = my_cnt[i];
            hist[i] = c ? atomicAdd(&cursor[i], c) : 0u;
        }
    } else {
        for (int i = tid; i < nbk; i += kSortThreads) hist[i] = 0u;
    }
    __syncthreads();
    const int q0 = ck * sp.qc;
    const int nq = sp.Lq - q0 < sp.qc ? sp.Lq - q0 : sp.qc;
    const int n_items = nq * LP;
    const unsigned char *const mk = mask ? mask + (size_t)b * sp.S : nullptr;
    for (int it = tid; it < n_items; it += kSortThreads) {
        const int ql = (it * sp.magic_lp) >> 16, t = it - ql * LP;
        const int l = t / P;
        const unsigned q = (unsigned)(q0 + ql);
        const unsigned pm = ((unsigned)b * (unsigned)sp.Lq + q) * (unsigned)sp.M + (unsigned)m;
        const f32x2 xy = *reinterpret_cast<const f32x2 *>(loc + ((size_t)pm * LP + t) * 2);
        const int H = s_H[l], W = s_W[l];
        const Sample<float> s = sample_setup<float>(xy.x, xy.y, H, W);
        if (!s.gate) continue;
        const int h0 = s.h_low, w0 = s.w_low;
        const bool okh0 = h0 >= 0, okh1 = h0 + 1 <= H - 1, okw0 = w0 >= 0, okw1 = w0 + 1 <= W - 1;
        bool v[4] = {okh0 && okw0, okh0 && okw1, okh1 && okw0, okh1 && okw1};
        const int p00 = s_start[l] + h0 * W + w0;
        const int px[4] = {p00, p00 + 1, p00 + W, p00 + W + 1};
        if (mk != nullptr) {       // padded pixels: value reads as 0 and receives no gradient
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = v[c] && !mk[v[c] ? px[c] : 0];
        }
        if (!EMIT) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (v[c]) atomicAdd(&hist[px[c] >> kSortBPLog], 1u);
        } else {
            const float a = attn[(size_t)pm * LP + t];
            const float hh = 1.f - s.lh, hw = 1.f - s.lw;
            const float wk[4] = {hh * hw, hh * s.lw, s.lh * hw, s.lh * s.lw};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (v[c]) {
                    const unsigned pos = atomicAdd(&hist[px[c] >> kSortBPLog], 1u);
                    sp.rec[pos] = u32x2{(q << kSortBPLog) | ((unsigned)px[c] & (unsigned)(kSortBP - 1)), __float_as_uint(wk[c] * a)};
                }
            }
        }
    }
    if (!EMIT) {
        __syncthreads();
        for (int i = tid; i < nbk; i += kSortThreads) {
            const unsigned c = hist[i];
            my_cnt[i] = c;
            if (c) atomicAdd(&cursor[i], c);
        }
    }
}

// exclusive prefix over the workgroup (256 threads), the total to every thread
__device__ __forceinline__ unsigned sort_block_excl(unsigned x, unsigned *s_wave, unsigned &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned incl = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned y = __shfl_up(incl, o, 64);
        if (lane >= o) incl += y;
    }
    __syncthreads();
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    unsigned before = 0u, tot = 0u;
#pragma unroll
    for (int w = 0; w < kSortThreads / 64; ++w) {
        const unsigned v = s_wave[w];
        before += w < wave ? v : 0u;
        tot += v;
    }
    total = tot;
    return before + incl - x;
}

// ---- scan: per (batch, head) bucket totals -> first record of every bucket, and the gather's work list ----
__global__ __launch_bounds__(kSortThreads) void msda_bwd_sort_scan(const SortPlan sp) {
    __shared__ unsigned s_wave[kSortThreads / 64];
    const int bm = blockIdx.x, tid = threadIdx.x, nbk = sp.nbk;
    unsigned *const cursor = sp.cursor + (size_t)bm * nbk;
    u32x4 *const work = sp.work + (size_t)bm * sp.max_items;
    // a thread owns `per` consecutive buckets
    const int per = (nbk + kSortThreads - 1) / kSortThreads;
    const int k0 = tid * per, k1 = k0 + per < nbk ? k0 + per : nbk;
    unsigned recs = 0u, items = 0u;
    for (int k = k0; k < k1; ++k) {
        const unsigned t = cursor[k];
        recs += t;
        items += (t + (unsigned)kSortSlice - 1u) / (unsigned)kSortSlice;
    }
    unsigned tot_recs, tot_items;
    unsigned start = sort_block_excl(recs, s_wave, tot_recs) + (unsigned)bm * sp.cap;
    unsigned wi = sort_block_excl(items, s_wave, tot_items);
    for (int k = k0; k < k1; ++k) {
        const unsigned t = cursor[k];
        cursor[k] = start;
        const unsigned ns = (t + (unsigned)kSortSlice - 1u) / (unsigned)kSortSlice;
        if (ns) {
            const unsigned len = (t + ns - 1u) / ns;        // equal slices, none above kSortSlice
            for (unsigned s = 0; s < ns; ++s) {
                const unsigned r0 = s * len, n = t - r0 < len ? t - r0 : len;
                work[wi++] = u32x4{(unsigned)k, start + r0, n, ns > 1u ? 1u : 0u};
            }
        }
        start += t;
    }
    if (tid == 0) sp.nwork[bm] = tot_items;
}

// ---- gather: one work item (a bucket, or a slice of a heavy one) per workgroup ----
template <typename TV>
__global__ __launch_bounds__(kSortThreads, 4) void msda_bwd_sort_gather(const TV *__restrict__ grad_out,
                                                                         float *__restrict__ grad_value,
                                                                         const SortPlan sp, unsigned go_bytes,
                                                                         unsigned gv_bytes) {
    constexpr bool kB16 = sizeof(TV) == 2;
    constexpr unsigned ROWB = 32u * (unsigned)sizeof(TV);
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    u32x2 *const E = reinterpret_cast<u32x2 *>(s_dyn);                                      // kSortSlice + 1 sorted entries
    unsigned *const CNT = reinterpret_cast<unsigned *>(s_dyn + (kSortSlice + 2) * 8);      // 64 counters
    unsigned *const START = CNT + kSortBP;                                                  // 64 starts
    const int tid = threadIdx.x, lane = tid & 63;
    // block -> (head, batch, item): head-major, one head per XCD
    const int per = sp.N * sp.max_items;
    const int chunk8 = (int)(gridDim.x >> 3);
    const int sw = (int)(blockIdx.x & 7) * chunk8 + (int)(blockIdx.x >> 3);
    if (sw >= per * sp.M) return;
    const int m = sw / per, rem = sw - m * per;
    const int b = rem / sp.max_items, item = rem - b * sp.max_items;
    const int bm = b * sp.M + m;
    if ((unsigned)item >= sp.nwork[bm]) return;
    const u32x4 w = sp.work[(size_t)bm * sp.max_items + item];
    const unsigned bucket = w.x, r0 = w.y, n = w.z;
    const bool sliced = w.w != 0u;
    if (tid < kSortBP) CNT[tid] = 0u;
    // the slice's records, all requested before the first is used
    u32x2 r[kSortRecPerThread];
#pragma unroll
    for (int k = 0; k < kSortRecPerThread; ++k) {
        const unsigned i = (unsigned)tid + (unsigned)k * kSortThreads;
        r[k] = sp.rec[r0 + (i < n ? i : 0u)];
    }
    __syncthreads();
    unsigned tk[kSortRecPerThread];
#pragma unroll
    for (int k = 0; k < kSortRecPerThread; ++k) {
        const unsigned i = (unsigned)tid + (unsigned)k * kSortThreads;
        tk[k] = i < n ? atomicAdd(&CNT[r[k].x & (unsigned)(kSortBP - 1)], 1u) : 0u;
    }
    __syncthreads();
    if (tid < 64) {
        const unsigned c = CNT[tid];
        unsigned incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned y = __shfl_up(incl, o, 64);
            if (lane >= o) incl += y;
        }
        START[tid] = incl - c;
        if (tid == 0) E[n] = u32x2{kOobOffset, 0u};        // the terminator: reads 0, weight 0
    }
    __syncthreads();
    const unsigned row_base = ((unsigned)b * (unsigned)sp.Lq * (unsigned)sp.M + (unsigned)m) * ROWB;
    const unsigned q_stride = (unsigned)sp.M * ROWB;
#pragma unroll
    for (int k = 0; k < kSortRecPerThread; ++k) {
        const unsigned i = (unsigned)tid + (unsigned)k * kSortThreads;
        if (i < n) {
            const unsigned d = r[k].x & (unsigned)(kSortBP - 1);
            E[START[d] + tk[k]] = u32x2{row_base + (r[k].x >> kSortBPLog) * q_stride, r[k].y};
        }
    }
    __syncthreads();

    // four lanes x eight channels per destination row
    const int grp = tid >> 2, j4 = tid & 3;
    const __amdgpu_buffer_rsrc_t gor = make_rsrc(grad_out, go_bytes);
    const unsigned cnt = CNT[grp];
    unsigned pe = START[grp];
    const unsigned lim = pe + cnt;
    // trip count: the longest list among this wavefront's 16 rows
    unsigned nm = cnt;
    nm = max(nm, BINS_DPP_U(nm, 0x124));      // row_ror:4
    nm = max(nm, BINS_DPP_U(nm, 0x128));      // row_ror:8
    const unsigned nmax = max(max((unsigned)__builtin_amdgcn_readlane((int)nm, 0), (unsigned)__builtin_amdgcn_readlane((int)nm, 16)),
                              max((unsigned)__builtin_amdgcn_readlane((int)nm, 32), (unsigned)__builtin_amdgcn_readlane((int)nm, 48)));
    // this lane's two 16-byte pieces of a row: fp32 rows: chunks j4 and j4 + 4 (a quad's load covers 64 contiguous
    // bytes); bf16 rows: ONE 16-byte piece = channels 8 j4 .. 8 j4 + 7
    const unsigned ca = (unsigned)j4 * 16u;
    f32x4 acc_a = f32x4{0.f, 0.f, 0.f, 0.f}, acc_b = f32x4{0.f, 0.f, 0.f, 0.f};
    for (unsigned i = 0; i < nmax; i += 4) {
        u32x2 e[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) e[k] = E[pe + (unsigned)k < lim ? pe + (unsigned)k : n];
        pe += 4u;
        f32x4 x[4], y[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (kB16) {
                const u32x4 u = buf_load_u4(gor, e[k].x + ca);
                x[k] = f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                             __uint_as_float(u.y & 0xffff0000u)};
                y[k] = f32x4{__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16),
                             __uint_as_float(u.w & 0xffff0000u)};
            } else {
                x[k] = buf_load_f4(gor, e[k].x + ca);
                y[k] = buf_load_f4(gor, e[k].x + ca + 64u);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float wk = __uint_as_float(e[k].y);
            acc_a += wk * x[k];
            acc_b += wk * y[k];
        }
    }
    const unsigned pix = (bucket << kSortBPLog) + (unsigned)grp;
    if (cnt == 0u || pix >= (unsigned)sp.S) return;
    // fp32 grad_value row: the bytes of this lane's two pieces
    const unsigned oa = kB16 ? (unsigned)j4 * 32u : ca, ob = kB16 ? (unsigned)j4 * 32u + 16u : ca + 64u;
    const unsigned goff = (((unsigned)b * (unsigned)sp.S + pix) * (unsigned)sp.M + (unsigned)m) * 128u;
    if (!sliced) {
        float *const dst = grad_value + (goff >> 2);
        *reinterpret_cast<f32x4 *>(dst + (oa >> 2)) = acc_a;
        *reinterpret_cast<f32x4 *>(dst + (ob >> 2)) = acc_b;
    } else {
        const __amdgpu_buffer_rsrc_t gvr = make_rsrc(grad_value, gv_bytes);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc_a[i], gvr, (int)(goff + oa + (unsigned)i * 4u), 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc_b[i], gvr, (int)(goff + ob + (unsigned)i * 4u), 0, 0);
        }
    }
}
