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
//   dots    (msda_bwd_sort_dots) per (batch, head, chunk of queries): grad_loc / grad_attn of the chunk -- the row phase of
//           msda_bwd_d32_bins without a region -- and, on the side, the chunk's COUNT: one lane per (query, point) does
//           the sampling arithmetic; the destination rows of a (batch, head) are cut into BUCKETS of 64 consecutive
//           pixels, the chunk histograms its corners per bucket in LDS, stores the histogram (its own row of a count
//           matrix) and adds it to the bucket totals (one fire-and-forget integer atomic per non-empty bucket);
//   scan    per (batch, head): bucket totals -> bucket starts (exclusive prefix; every (batch, head) owns a fixed
//           segment of the record array, so no prefix crosses workgroups), and the gather's WORK LIST: a bucket with
//           more records than one workgroup sorts in LDS is cut into equal slices;
//   emit    the item arithmetic again, now writing: a workgroup reserves its share of every bucket with ONE returning
//           atomic per bucket (its chunks' counts from the matrix), then every corner pair takes two LDS tickets and
//           leaves as one 16-byte store of two 8-byte records {query << 6 | pixel inside the bucket, w_corner * attn}
//           -- 91 MB for the encoder call;
//   gather  a workgroup per work item: counting sort of the slice's records by destination pixel in LDS (64 counters,
//           integer tickets), then four lanes x eight channels per destination row walk the row's list:
//           acc += w * grad_out[query, head, :] (16-byte loads, one head per XCD so that its 2.9 MB grad_out slab stays
//           in that XCD's L2), fp32 accumulation in registers, and ONE plain 128-byte store per row;
//   reduce  the rows of sliced buckets (coarse pyramid levels, where thousands of points share a pixel): their slices'
//           partial rows, kept in the scratch, added in slice order.  No float atomic anywhere in the call.
//
// The fused call: grad_loc / grad_attn go into the columns of grad_proj; for the encoder's rows (L * P = 16, 2-d reference
// points) the dots and emit kernels form the softmax weights themselves and the dots kernel applies the softmax Jacobian,
// otherwise the side kernels of msda_fused_side.h materialise the weights and finish the Jacobians; when gradients of the
// reference points are wanted msda_bwd_d32_rows runs without its atomics.  Non-finite gradients propagate through fp32
// arithmetic like the reference's atomicAdd.
#pragma once

constexpr int kSortBPLog = 6, kSortBP = 1 << kSortBPLog;      // destination pixels per bucket
constexpr int kSortThreads = 256;
#ifndef MSDA_SORT_SLICE
#define MSDA_SORT_SLICE 4608
#endif
#ifndef MSDA_SORT_UN
#define MSDA_SORT_UN 8
#endif
constexpr int kSortSlice = MSDA_SORT_SLICE;   // records one gathering workgroup sorts (4608: 36 KB of LDS, four workgroups per CU)
constexpr int kSortRecPerThread = kSortSlice / kSortThreads;
constexpr int kSortMaxBuckets = 8192;    // per (batch, head): the histogram lives in LDS (32 KB) next to the dots kernel's chunk

struct SortPlan {
    int N, S, M, L, Lq, P;
    int qc;                  // queries per dots-and-count workgroup
    int nchunk;              // ... workgroups per (batch, head): ceil(Lq / qc)
    int emult;               // chunks per EMIT workgroup (longer runs per bucket, fewer reservations)
    int nbk;                 // buckets per (batch, head): ceil(S / 64)
    int magic_lp;            // (i * magic_lp) >> 16 == i / (L * P) for i < qc * L * P
    unsigned cap;            // records per (batch, head) segment: Lq * L * P * 4
    int max_items;           // work items per (batch, head), upper bound
    unsigned *cnt;           // [N * M][nchunk][nbk] corner counts per counting workgroup and bucket
    unsigned *cursor;        // [N * M][nbk] bucket totals (count), then the next free record of the bucket (scan, emit)
    unsigned *nwork;         // [N * M]
    u32x4 *work;             // [N * M][max_items] {bucket, first record, records, sliced}
    u32x2 *red;              // [N * M][nbk] {first work item, slices} of every bucket (slices > 1: its rows are reduced)
    float *part;             // [N * M][max_items][64 rows][32] partial row sums of the slices of sliced buckets
    u32x2 *rec;              // [N * M][cap]
    size_t bytes;            // of the whole scratch block
};

// Layout of the scratch block; false when the call does not fit the kernels' 32-bit arithmetic.
inline bool make_sort_plan(SortPlan &sp, int N, int S, int M, int L, int Lq, int P, size_t elem_bytes, void *workspace,
                           int qc_opt = 0, int emult_opt = 0) {
    memset(&sp, 0, sizeof(sp));
    const long LP = (long)L * P;
    if (LP < 1 || LP > kRowsMaxLP || N < 1 || Lq < 1 || Lq >= (1 << 26)) return false;
    const long nbk = ((long)S + kSortBP - 1) >> kSortBPLog;
    if (nbk > kSortMaxBuckets) return false;
    const long cap = (long)Lq * LP * 4, total = cap * N * M;
    if (total >= (1L << 31) || (long)N * Lq * M * 32 * (long)elem_bytes >= 0x7fffff00L) return false;
    // a chunk's grad_out rows (144 B each), item records (16 + 4 B) and its histogram live in LDS next to four other
    // workgroups': 64 queries at L * P = 16 (28 KB); the emit kernel takes two chunks per workgroup at that size
    int qc = 64;
    while (qc > 8 && (size_t)(qc + 1) * 144 + (size_t)qc * LP * 20 > 28 * 1024) qc >>= 1;
    if (qc_opt >= 8 && qc_opt <= 128 && (qc_opt & (qc_opt - 1)) == 0 && (size_t)(qc_opt + 1) * 144 + (size_t)qc_opt * LP * 20 <= 56 * 1024)
        qc = qc_opt;       // ("bwd_sort_qc": measurements)
    int emult = (long)N * M * (((Lq + qc - 1) / qc + 1) / 2) >= 1024 ? 2 : 1;
    if (emult_opt >= 1 && emult_opt <= 8) emult = emult_opt;
    const int items = emult * qc * (int)LP;
    const int magic = 65536 / (int)LP + 1;
    for (int i = 0; i < items; ++i)
        if (((i * magic) >> 16) != i / (int)LP) return false;
    sp.N = N; sp.S = S; sp.M = M; sp.L = L; sp.Lq = Lq; sp.P = P;
    sp.qc = qc; sp.nchunk = (Lq + qc - 1) / qc; sp.nbk = (int)nbk; sp.magic_lp = magic; sp.cap = (unsigned)cap;
    sp.emult = emult;
    sp.max_items = (int)(cap / kSortSlice + nbk);
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t nbm = (size_t)N * M;
    size_t o = 0;
    unsigned char *base = reinterpret_cast<unsigned char *>(workspace);
    sp.cursor = reinterpret_cast<unsigned *>(base + o); o += up(nbm * nbk * 4);
    sp.nwork = reinterpret_cast<unsigned *>(base + o); o += up(nbm * 4);
    sp.cnt = reinterpret_cast<unsigned *>(base + o); o += up(nbm * sp.nchunk * nbk * 4);
    sp.work = reinterpret_cast<u32x4 *>(base + o); o += up(nbm * sp.max_items * 16);
    sp.red = reinterpret_cast<u32x2 *>(base + o); o += up(nbm * nbk * 8);
    sp.part = reinterpret_cast<float *>(base + o); o += up(nbm * sp.max_items * (size_t)kSortBP * 128);
    sp.rec = reinterpret_cast<u32x2 *>(base + o); o += up(nbm * cap * 8);
    sp.bytes = o;
    return true;
}

// Where a point's sampling location comes from: the operator's `loc`, or (fused_loc, the slim split fused backward) the
// module's own arithmetic on the raw projection row and the reference points -- the bits msda_fused_points_f32 exposes.
// The attention weight is materialised either way (src.attn: the operator's, or the workspace copy of the softmax).
struct SortRaw {
    f32x2 off, r01, r23;     // plain: off = the location itself
    float a;
};
__device__ __forceinline__ SortRaw sort_load_raw(const PointSrc &src, int fused_loc, unsigned qrow, unsigned pm, unsigned m,
                                                 unsigned L, unsigned LP, unsigned t, unsigned l, bool soft16) {
    SortRaw r;
    // (32-bit element indices: check_dims keeps every tensor below 2^31 elements and qrow * proj_stride below 1.5 * 2^31)
    const float *const p_raw = fused_loc ? src.proj + (qrow * (unsigned)src.proj_stride + (m * LP + t) * 2u)
                                         : src.loc + (pm * LP + t) * 2u;
    const float *const rp = fused_loc ? src.ref + (qrow * L + l) * (unsigned)src.ref_dim : p_raw;
    r.off = *reinterpret_cast<const f32x2 *>(p_raw);
    r.r01 = *reinterpret_cast<const f32x2 *>(rp);
    r.r23 = *reinterpret_cast<const f32x2 *>(fused_loc && src.ref_dim != 2 ? rp + 2 : p_raw);
    // soft16: the row's logit -- its sixteen lanes turn it into the softmax weight (sort_row16_softmax)
    r.a = soft16 ? src.proj[qrow * (unsigned)src.proj_stride + (unsigned)src.n_off + m * LP + t] : src.attn[pm * LP + t];
    return r;
}
__device__ __forceinline__ f32x2 sort_location(const SortRaw &r, const PointSrc &src, int fused_loc, int P, int H, int W) {
    return fused_loc ? fused_location_from(r.off, r.r01, r.r23, src.ref_dim, P, H, W) : r.off;
}

// The softmax weight of a point from its logit, by the sixteen lanes (one DPP row) that hold the L * P = 16 logits of a
// (query, head) row: msda_common.h's arithmetic (sm_exp / sm_rcp, adjacent-pair tree = the butterflies t ^ 1, 2, 4, 8),
// the bits of the windowed forward and of msda_fused_attn16_rows_kernel.
__device__ __forceinline__ float sort_row16_softmax(float lg) {
    float mx = lg;
    mx = fmaxf(mx, MSDA_DPP(mx, 0xB1));
    mx = fmaxf(mx, MSDA_DPP(mx, 0x4E));
    mx = fmaxf(mx, MSDA_DPP(mx, 0x141));
    mx = fmaxf(mx, MSDA_DPP(mx, 0x140));
    const float e = sm_exp(lg, mx);
    float sum = e + MSDA_DPP(e, 0xB1);           // t ^ 1
    sum += MSDA_DPP(sum, 0x4E);                  // t ^ 2
    sum += MSDA_DPP(sum, 0x141);                 // the other quad of the eight (all four lanes of a quad hold one value)
    sum += MSDA_DPP(sum, 0x140);                 // the other eight
    return e * sm_rcp(sum);
}

// ---- dots + count: grad_loc / grad_attn of a chunk of queries of one head, and the chunk's corner histogram ----
// The row phase of msda_bwd_d32_bins without its region.  A workgroup owns (batch, head, chunk of `qc` queries):
//   * the chunk's grad_out rows are staged in LDS once (fp32, 144-byte pitch);
//   * one lane per (query, point) ITEM does the scalar work once -- location (the operator's, or fused_loc: the module's
//     arithmetic on the raw projection), sample arithmetic, corner validity / padding mask -- leaves a 16-byte record
//     {value byte offset of the top-left corner, lh, lw, attention} + a flag byte in LDS, and counts the item's valid
//     corners into the chunk's histogram over the destination buckets (what the emit kernel reserves by, and -- added to
//     the bucket totals -- what the scan turns into bucket starts);
//   * four lanes x eight channels per item then compute the four corner dot products d_k = <grad_out_row, v_k> (16-byte
//     loads of the corner rows, two DPP steps) from which all three gradients of the point derive: grad_attn =
//     sum_k w_k d_k, d/dx = a W (hh (d1 - d0) + lh (d3 - d2)), d/dy = a H (hw (d2 - d0) + lw (d3 - d1)); lanes 0 / 1 / 2
//     of the quad write them.
// A first version without LDS -- every lane of a quad redoing the item's scalar work and loading its own copy of the
// grad_out row and of the inputs, one task per round trip -- took 98 / 120 us (near / uniform locations) next to a
// separate 15 us counting kernel: neither VALU- nor tail-bound (315 instead of 534 static VALU instructions and an
// occupancy-sized grid changed nothing) but by the redundant small loads in front of every task's corner rows
// (profiles/r06_sorted_ab.txt).
// Workgroups are numbered head-major, one head per XCD: the head's slab of `value` stays in that L2.
// grad_proj != null: the split fused backward -- the results go to the offset / logit columns of grad_proj
// (offsets_done: as final offset gradients, d loc / d offset = 1 / (W, H) for 2-d reference points) and the finishing
// kernels of msda_fused_side.h apply the remaining Jacobians in place.
struct SortDotsLds {
    unsigned o_rec, o_fl, o_hist, bytes;      // (grad_out rows at 0)
};
inline SortDotsLds sort_dots_lds(int qc, int LP, int nbk) {
    SortDotsLds d;
    auto up16 = [](unsigned x) { return (x + 15u) & ~15u; };
    d.o_rec = up16((unsigned)(qc + 1) * kBinsGRow);
    d.o_fl = d.o_rec + (unsigned)(qc * LP) * 16u;
    d.o_hist = up16(d.o_fl + (unsigned)(qc * LP) * 4u);
    d.bytes = d.o_hist + (unsigned)nbk * 4u;
    return d;
}

template <typename TV>
__global__ __launch_bounds__(kSortThreads) void msda_bwd_sort_dots(
    const TV *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lstart,
    const PointSrc src, int fused_loc, int offsets_done, int soft16, const TV *__restrict__ grad_out,
    float *__restrict__ grad_loc, float *__restrict__ grad_attn, float *__restrict__ grad_proj, const SortPlan sp,
    const SortDotsLds ld, unsigned value_bytes) {
    constexpr bool kB16 = sizeof(TV) == 2;
    constexpr unsigned ROWB = 32u * (unsigned)sizeof(TV);
    constexpr unsigned kChunkDelta = kB16 ? 16u : 64u;      // between a lane's two 16-byte chunks of a staged grad_out row
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    __shared__ int s_H[kMaxLevels], s_W[kMaxLevels], s_start[kMaxLevels];
    unsigned char *const G = s_dyn;
    u32x4 *const R = reinterpret_cast<u32x4 *>(s_dyn + ld.o_rec);
    unsigned *const FL = reinterpret_cast<unsigned *>(s_dyn + ld.o_fl);
    unsigned *const hist = reinterpret_cast<unsigned *>(s_dyn + ld.o_hist);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int per = sp.N * sp.nchunk;                   // workgroups per head
    const int chunk8 = (int)(gridDim.x >> 3);
    const int sw = (int)(blockIdx.x & 7) * chunk8 + (int)(blockIdx.x >> 3);
    if (sw >= per * sp.M) return;
    const int m = sw / per, rem = sw - m * per;
    const int b = rem / sp.nchunk, ck = rem - b * sp.nchunk;
    const int bm = b * sp.M + m;
    const int L = sp.L, P = sp.P, LP = L * P, M = sp.M, nbk = sp.nbk;
    if (tid < L) {
        s_H[tid] = (int)shapes[2 * tid];
        s_W[tid] = (int)shapes[2 * tid + 1];
        s_start[tid] = (int)lstart[tid];
    }
    for (int i = tid; i < nbk; i += kSortThreads) hist[i] = 0u;
    const int q0 = ck * sp.qc;
    const int nq = sp.Lq - q0 < sp.qc ? sp.Lq - q0 : sp.qc;
    const int n_items = nq * LP;
    const unsigned qrow0 = (unsigned)b * (unsigned)sp.Lq + (unsigned)q0;
    // ---- phase 0: the chunk's grad_out rows -> LDS (fp32); row nq is the zero row of items that do not exist ----
    for (int idx = tid; idx < (nq + 1) * 8; idx += kSortThreads) {
        const int r = idx >> 3;
        const unsigned pm = (qrow0 + (unsigned)(r < nq ? r : 0)) * (unsigned)M + (unsigned)m;
        f32x4 g = bins_load_g4<TV>(grad_out + (pm * 32u + (unsigned)((idx & 7) * 4)));
        if (r >= nq) g = f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4 *>(G + (unsigned)r * kBinsGRow + (unsigned)(idx & 7) * 16u) = g;
    }
    __syncthreads();      // (level tables, cleared histogram)
    // ---- phase 1: one lane per item; two items per iteration, their inputs requested together ----
    const unsigned char *const mk = src.mask ? src.mask + (size_t)b * sp.S : nullptr;
    for (int it0 = tid; it0 < n_items; it0 += 2 * kSortThreads) {
        SortRaw raw[2];
        unsigned tt[2], ll[2];
        bool ok[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int it = it0 + u * kSortThreads;
            ok[u] = it < n_items;
            const int itc = ok[u] ? it : it0;
            const int ql = (itc * sp.magic_lp) >> 16;
            tt[u] = (unsigned)(itc - ql * LP);
            ll[u] = tt[u] / (unsigned)P;
            const unsigned qrow = qrow0 + (unsigned)ql;
            raw[u] = sort_load_raw(src, fused_loc, qrow, qrow * (unsigned)M + (unsigned)m, (unsigned)m, (unsigned)L,
                                   (unsigned)LP, tt[u], ll[u], soft16 != 0);
        }
        // (soft16: L * P = 16 -- the sixteen lanes of a DPP row hold one (query, head) row's logits, all of them existing
        //  or none; the weights never pass through HBM)
        if (soft16) {
            raw[0].a = sort_row16_softmax(raw[0].a);
            raw[1].a = sort_row16_softmax(raw[1].a);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!ok[u]) continue;
            const unsigned l = ll[u];
            const int H = s_H[l], W = s_W[l];
            const f32x2 xy = sort_location(raw[u], src, fused_loc, P, H, W);
            const Sample<float> s = sample_setup<float>(xy.x, xy.y, H, W);
            const bool gate = s.gate;
            const int h0 = s.h_low, w0 = s.w_low;
            const bool okh0 = gate && h0 >= 0, okh1 = gate && h0 + 1 <= H - 1, okw0 = w0 >= 0, okw1 = w0 + 1 <= W - 1;
            bool v[4] = {okh0 && okw0, okh0 && okw1, okh1 && okw0, okh1 && okw1};
            const int p00 = s_start[l] + h0 * W + w0;
            const int px[4] = {p00, p00 + 1, p00 + W, p00 + W + 1};
            if (mk != nullptr) {       // padded pixels: value reads as 0 and receives no gradient
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = v[c] && !mk[v[c] ? px[c] : 0];
            }
            // the histogram the emit kernel's tickets will follow: a pixel row's two corners count together unless
            // they straddle a bucket boundary (msda_bwd_sort_emit draws its tickets by the same rule)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const int c0 = 2 * pr, c1 = c0 + 1;
                const bool both = v[c0] && v[c1] && ((unsigned)px[c0] & (unsigned)(kSortBP - 1)) != (unsigned)(kSortBP - 1);
                if (both) {
                    atomicAdd(&hist[px[c0] >> kSortBPLog], 2u);
                } else {
                    if (v[c0]) atomicAdd(&hist[px[c0] >> kSortBPLog], 1u);
                    if (v[c1]) atomicAdd(&hist[px[c1] >> kSortBPLog], 1u);
                }
            }
            u32x4 rec;
            rec.x = (((unsigned)b * (unsigned)sp.S + (unsigned)p00) * (unsigned)M + (unsigned)m) * ROWB;
            rec.y = __float_as_uint(gate ? s.lh : 0.f);
            rec.z = __float_as_uint(gate ? s.lw : 0.f);
            rec.w = __float_as_uint(raw[u].a);        // (the weight itself: the softmax Jacobian wants it for gated-off points too)
            const int it = it0 + u * kSortThreads;
            R[it] = rec;
            FL[it] = (v[0] ? 1u : 0u) | (v[1] ? 2u : 0u) | (v[2] ? 4u : 0u) | (v[3] ? 8u : 0u) | (gate ? 16u : 0u);
        }
    }
    __syncthreads();
    // the chunk's histogram: its row of the count matrix, and into the bucket totals
    {
        unsigned *const my_cnt = sp.cnt + ((size_t)bm * sp.nchunk + ck) * nbk;
        unsigned *const cursor = sp.cursor + (size_t)bm * nbk;
        for (int i = tid; i < nbk; i += kSortThreads) {
            const unsigned c = hist[i];
            my_cnt[i] = c;
            if (c) atomicAdd(&cursor[i], c);
        }
    }
    // ---- phase 2: four lanes x eight channels per item, 16 items per wavefront step ----
    const int grp = lane >> 2, j4 = lane & 3;
    const unsigned ch_a = kB16 ? 2u * (unsigned)j4 : (unsigned)j4;      // this lane's two 16-byte chunks of a staged row
    const __amdgpu_buffer_rsrc_t vr = make_rsrc(value, value_bytes);
    const unsigned ps = (unsigned)M * ROWB;
    const bool role_y = j4 == 1, role_a = j4 == 2;
    const bool split = grad_proj != nullptr;
    // the softmax Jacobian in place of the finishing kernel: L * P = 16, so a wavefront step IS one (query, head) row and
    // its sixteen quads hold the row's d/d(attention) -- grad_logit_t = a_t (ga_t - sum_j a_j ga_j)
    const bool fuse_jac = split && soft16 != 0;
    struct Step {
        u32x4 rec;
        unsigned fl, t, l;
        int r;
        bool vi;
        u32x4 ua[4], ub[4];
    };
    auto issue = [&](int st) -> Step {
        Step x;
        const int it = st * 16 + grp;
        x.vi = it < n_items;
        const int itc = x.vi ? it : n_items - 1;
        x.rec = R[itc];
        x.fl = x.vi ? FL[itc] : 0u;
        x.r = (itc * sp.magic_lp) >> 16;
        x.t = (unsigned)(itc - x.r * LP);
        x.l = x.t / (unsigned)P;
        const unsigned wps = (unsigned)s_W[x.l] * ps;
        const unsigned base = x.rec.x + (unsigned)j4 * 16u;
        const unsigned off[4] = {(x.fl & 1u) ? base : kOobOffset, (x.fl & 2u) ? base + ps : kOobOffset,
                                 (x.fl & 4u) ? base + wps : kOobOffset, (x.fl & 8u) ? base + wps + ps : kOobOffset};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            x.ua[k] = buf_load_u4(vr, off[k]);
            if (!kB16) x.ub[k] = __builtin_amdgcn_raw_buffer_load_b128(vr, (int)off[k], 64, 0);
        }
        return x;
    };
    const int n_steps = (n_items + 15) >> 4;
    if (wave >= n_steps) return;
    Step cur = issue(wave);
    for (int st = wave; st < n_steps; st += kSortThreads / 64) {
        // the next step's corner rows travel while this step's are used (a wavefront's steps were a chain of round trips)
        const bool more = st + kSortThreads / 64 < n_steps;
        Step nxt;
        if (more) nxt = issue(st + kSortThreads / 64);
        const u32x4 rec = cur.rec;
        const unsigned fl = cur.fl, t = cur.t, l = cur.l;
        const int r = cur.r;
        const bool vi = cur.vi;
        const unsigned char *const grow = G + (unsigned)(vi ? r : nq) * kBinsGRow + ch_a * 16u;
        const f32x4 ga = *reinterpret_cast<const f32x4 *>(grow);
        const f32x4 gb = *reinterpret_cast<const f32x4 *>(grow + kChunkDelta);
        const int W = s_W[l], H = s_H[l];
        float d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f32x4 va, vb;
            if (kB16) {
                const u32x4 u = cur.ua[k];
                va = f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                           __uint_as_float(u.y & 0xffff0000u)};
                vb = f32x4{__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16),
                           __uint_as_float(u.w & 0xffff0000u)};
            } else {
                va = __builtin_bit_cast(f32x4, cur.ua[k]);
                vb = __builtin_bit_cast(f32x4, cur.ub[k]);
            }
            d[k] = bins_dot8(ga, va, gb, vb);
            MSDA_QUAD_SUM(d[k]);
        }
        // a gated-off point contributes exactly nothing, also for non-finite gradients
        const bool gate = (fl & 16u) != 0u;
        const float lh = __uint_as_float(rec.y), lw = __uint_as_float(rec.z), a_raw = __uint_as_float(rec.w);
        const float a = gate ? a_raw : 0.f;
        const float hh = 1.f - lh, hw = 1.f - lw;
        const float pp = role_y ? hw : hh, qq = role_y ? lw : lh;
        const float d_a = role_y ? d[2] : d[1], d_c = role_y ? d[1] : d[2];
        const float size_r = offsets_done ? 1.f : (role_y ? (float)H : (float)W);
        const float r_loc = gate ? (a * size_r) * (pp * (d_a - d[0]) + qq * (d[3] - d_c)) : 0.f;
        float r_att = gate ? (hh * hw) * d[0] + (hh * lw) * d[1] + (lh * hw) * d[2] + (lh * lw) * d[3] : 0.f;
        if (fuse_jac) {
            // sum over the row's sixteen points = over the wavefront's sixteen quads (the four lanes of a quad agree);
            // associated like msda_fused_finish16_rows_kernel: (t, t + 8), (t, t + 4), then (0 + 2) + (1 + 3)
            float dt = a_raw * r_att;
            dt += __shfl_xor(dt, 32, 64);       // t ^ 8
            dt += __shfl_xor(dt, 16, 64);       // t ^ 4
            // now every lane holds D[t & 3] (D_i = (d_i + d_i+8) + (d_i+4 + d_i+12)); the other three, inside the DPP row:
            const float o1 = MSDA_DPP(dt, 0x141);       // row_half_mirror: the quad (t & 3) ^ 1
            const float o2 = MSDA_DPP(dt, 0x140);       // row_mirror:      the quad (t & 3) ^ 3
            const float o3 = MSDA_DPP(o1, 0x140);       //                  the quad (t & 3) ^ 2
            // (D_0 + D_2) + (D_1 + D_3): this lane's pair is D_p + D_p^2, the other D_p^1 + D_p^3 -- sums commute
            const float dot = (dt + o3) + (o1 + o2);
            r_att = a_raw * (r_att - dot);
        }
        if (vi && j4 < 3) {
            const unsigned qrow = qrow0 + (unsigned)r, pm = qrow * (unsigned)M + (unsigned)m;
            float *dst;
            if (split)
                dst = grad_proj + (qrow * (unsigned)src.proj_stride +
                                   (role_a ? (unsigned)src.n_off + (unsigned)m * (unsigned)LP + t
                                           : ((unsigned)m * (unsigned)LP + t) * 2u + (unsigned)j4));
            else
                dst = role_a ? grad_attn + (pm * (unsigned)LP + t) : grad_loc + ((pm * (unsigned)LP + t) * 2u + (unsigned)j4);
            *dst = role_a ? r_att : r_loc;
        }
        if (!more) break;
        cur = nxt;
    }
}

// ---- emit: one lane per (query, point) of one head writes the point's corner records ----
__global__ __launch_bounds__(kSortThreads) void msda_bwd_sort_emit(const int64_t *__restrict__ shapes,
                                                                    const int64_t *__restrict__ lstart,
                                                                    const PointSrc src, int fused_loc, int soft16,
                                                                    const SortPlan sp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    __shared__ int s_H[kMaxLevels], s_W[kMaxLevels], s_start[kMaxLevels];
    unsigned *const hist = reinterpret_cast<unsigned *>(s_dyn);         // next free record of every bucket
    const int tid = threadIdx.x;
    // block -> (head, batch, emit chunk): head-major, one head per XCD like the gather (the records of a head are
    // written and read through the same L2).  An emit chunk is `emult` chunks of the dots kernel.
    const int nech = (sp.nchunk + sp.emult - 1) / sp.emult;
    const int per = sp.N * nech;
    const int chunk8 = (int)(gridDim.x >> 3);
    const int sw = (int)(blockIdx.x & 7) * chunk8 + (int)(blockIdx.x >> 3);
    if (sw >= per * sp.M) return;
    const int m = sw / per, rem = sw - m * per;
    const int b = rem / nech, eck = rem - b * nech;
    const int bm = b * sp.M + m;
    const int L = sp.L, P = sp.P, LP = L * P, nbk = sp.nbk;
    if (tid < L) {
        s_H[tid] = (int)shapes[2 * tid];
        s_W[tid] = (int)shapes[2 * tid + 1];
        s_start[tid] = (int)lstart[tid];
    }
    // this workgroup's share of every bucket: one returning atomic per non-empty bucket
    {
        const int c0 = eck * sp.emult, c1 = c0 + sp.emult < sp.nchunk ? c0 + sp.emult : sp.nchunk;
        const unsigned *const cnt0 = sp.cnt + ((size_t)bm * sp.nchunk + c0) * nbk;
        unsigned *const cursor = sp.cursor + (size_t)bm * nbk;
        for (int i = tid; i < nbk; i += kSortThreads) {
            unsigned c = 0u;
            for (int j = 0; j < c1 - c0; ++j) c += cnt0[(size_t)j * nbk + i];
            hist[i] = c ? atomicAdd(&cursor[i], c) : 0u;
        }
    }
    __syncthreads();
    const int q0 = eck * sp.emult * sp.qc;
    const int nq = sp.Lq - q0 < sp.emult * sp.qc ? sp.Lq - q0 : sp.emult * sp.qc;
    const int n_items = nq * LP;
    const unsigned char *const mk = src.mask ? src.mask + (size_t)b * sp.S : nullptr;
    // two items per lane and iteration: their inputs are requested together (a lane's chain is load -> ticket -> store)
    for (int it0 = tid; it0 < n_items; it0 += 2 * kSortThreads) {
        SortRaw raw[2];
        unsigned qq[2], ll[2];
        bool ok[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int it = it0 + u * kSortThreads;
            ok[u] = it < n_items;
            const unsigned itc = (unsigned)(ok[u] ? it : it0);
            const unsigned ql = itc / (unsigned)LP, t = itc - ql * (unsigned)LP;
            ll[u] = t / (unsigned)P;
            qq[u] = (unsigned)q0 + ql;
            const unsigned qrow = (unsigned)b * (unsigned)sp.Lq + qq[u];
            raw[u] = sort_load_raw(src, fused_loc, qrow, qrow * (unsigned)sp.M + (unsigned)m, (unsigned)m, (unsigned)L,
                                   (unsigned)LP, t, ll[u], soft16 != 0);
        }
        if (soft16) {       // (as in the dots kernel: the same function, the same bits)
            raw[0].a = sort_row16_softmax(raw[0].a);
            raw[1].a = sort_row16_softmax(raw[1].a);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const unsigned q = qq[u], l = ll[u];
            const int H = s_H[l], W = s_W[l];
            const f32x2 xy = sort_location(raw[u], src, fused_loc, P, H, W);
            const Sample<float> s = sample_setup<float>(xy.x, xy.y, H, W);
            if (!(s.gate && ok[u])) continue;
            const int h0 = s.h_low, w0 = s.w_low;
            const bool okh0 = h0 >= 0, okh1 = h0 + 1 <= H - 1, okw0 = w0 >= 0, okw1 = w0 + 1 <= W - 1;
            bool v[4] = {okh0 && okw0, okh0 && okw1, okh1 && okw0, okh1 && okw1};
            const int p00 = s_start[l] + h0 * W + w0;
            const int px[4] = {p00, p00 + 1, p00 + W, p00 + W + 1};
            if (mk != nullptr) {       // padded pixels: value reads as 0 and receives no gradient
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = v[c] && !mk[v[c] ? px[c] : 0];
            }
            // The two corners of a pixel row are neighbours: unless they straddle a bucket boundary they take TWO
            // consecutive tickets with one LDS atomic and leave as one 16-byte store (half the L2 write requests of the
            // emit, which is bound by them: 11.4 M scattered 8-byte stores = 72 us, profiles/r06_sorted_stats_v1.txt)
            const float a = raw[u].a;
            const float hh = 1.f - s.lh, hw = 1.f - s.lw;
            const float wk[4] = {(hh * hw) * a, (hh * s.lw) * a, (s.lh * hw) * a, (s.lh * s.lw) * a};
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const int c0 = 2 * pr, c1 = c0 + 1;
                const unsigned d0 = (unsigned)px[c0] & (unsigned)(kSortBP - 1), d1 = (unsigned)px[c1] & (unsigned)(kSortBP - 1);
                const bool both = v[c0] && v[c1] && d0 != (unsigned)(kSortBP - 1);
                if (both) {
                    const unsigned pos = atomicAdd(&hist[px[c0] >> kSortBPLog], 2u);
                    typedef u32x4 __attribute__((aligned(8))) u32x4_a8;
                    *reinterpret_cast<u32x4_a8 *>(sp.rec + pos) =
                        u32x4{(q << kSortBPLog) | d0, __float_as_uint(wk[c0]), (q << kSortBPLog) | d1, __float_as_uint(wk[c1])};
                } else {
                    if (v[c0]) {
                        const unsigned pos = atomicAdd(&hist[px[c0] >> kSortBPLog], 1u);
                        sp.rec[pos] = u32x2{(q << kSortBPLog) | d0, __float_as_uint(wk[c0])};
                    }
                    if (v[c1]) {
                        const unsigned pos = atomicAdd(&hist[px[c1] >> kSortBPLog], 1u);
                        sp.rec[pos] = u32x2{(q << kSortBPLog) | d1, __float_as_uint(wk[c1])};
                    }
                }
            }
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
        sp.red[(size_t)bm * nbk + k] = u32x2{wi, ns};
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
                                                                         const SortPlan sp, unsigned go_bytes) {
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
    // eight entries (sixteen 16-byte loads) in flight per lane; the next batch's entries are read from LDS while this
    // batch's rows travel (a workgroup's list walk is a chain of round trips: 4 per batch measured 117 us at uniform)
    constexpr int UN = MSDA_SORT_UN;
    u32x2 e[UN];
#pragma unroll
    for (int k = 0; k < UN; ++k) e[k] = E[pe + (unsigned)k < lim ? pe + (unsigned)k : n];
    for (unsigned i = 0; i < nmax; i += UN) {
        u32x4 x[UN], y[UN];
#pragma unroll
        for (int k = 0; k < UN; ++k) {
            x[k] = buf_load_u4(gor, e[k].x + ca);
            if (!kB16) y[k] = buf_load_u4(gor, e[k].x + ca + 64u);
        }
        float wk[UN];
#pragma unroll
        for (int k = 0; k < UN; ++k) wk[k] = __uint_as_float(e[k].y);
        pe += (unsigned)UN;
#pragma unroll
        for (int k = 0; k < UN; ++k) e[k] = E[pe + (unsigned)k < lim ? pe + (unsigned)k : n];
        __builtin_amdgcn_sched_barrier(0);      // (keep the requests above the FMAs)
#pragma unroll
        for (int k = 0; k < UN; ++k) {
            f32x4 xa, xb;
            if (kB16) {
                xa = f32x4{__uint_as_float(x[k].x << 16), __uint_as_float(x[k].x & 0xffff0000u), __uint_as_float(x[k].y << 16),
                           __uint_as_float(x[k].y & 0xffff0000u)};
                xb = f32x4{__uint_as_float(x[k].z << 16), __uint_as_float(x[k].z & 0xffff0000u), __uint_as_float(x[k].w << 16),
                           __uint_as_float(x[k].w & 0xffff0000u)};
            } else {
                xa = __builtin_bit_cast(f32x4, x[k]);
                xb = __builtin_bit_cast(f32x4, y[k]);
            }
            acc_a += wk[k] * xa;
            acc_b += wk[k] * xb;
        }
    }
    const unsigned pix = (bucket << kSortBPLog) + (unsigned)grp;
    // fp32 grad_value row: the bytes of this lane's two pieces
    const unsigned oa = kB16 ? (unsigned)j4 * 32u : ca, ob = kB16 ? (unsigned)j4 * 32u + 16u : ca + 64u;
    if (sliced) {
        // a slice of a heavy bucket (the coarse pyramid levels: thousands of points share a pixel): its partial row sums go
        // to the scratch, msda_bwd_sort_reduce adds the bucket's slices in a fixed order.  (Float atomics here -- 8 per lane
        // -- were 5 M lane-atomics and 68 of the gather's 90 MB of writes at the encoder shape, profiles/r06_pmc_bwd_sorted_*.txt)
        float *const dst = sp.part + (((size_t)bm * sp.max_items + item) * kSortBP + (unsigned)grp) * 32u;
        *reinterpret_cast<f32x4 *>(dst + (oa >> 2)) = acc_a;
        *reinterpret_cast<f32x4 *>(dst + (ob >> 2)) = acc_b;
        return;
    }
    if (cnt == 0u || pix >= (unsigned)sp.S) return;
    const unsigned goff = (((unsigned)b * (unsigned)sp.S + pix) * (unsigned)sp.M + (unsigned)m) * 128u;
    float *const dst = grad_value + (goff >> 2);
    *reinterpret_cast<f32x4 *>(dst + (oa >> 2)) = acc_a;
    *reinterpret_cast<f32x4 *>(dst + (ob >> 2)) = acc_b;
}

// ---- reduce: the rows of sliced buckets = the sum of their slices' partial rows, slice by slice ----
__global__ __launch_bounds__(kSortThreads) void msda_bwd_sort_reduce(float *__restrict__ grad_value, const SortPlan sp) {
    const int per = sp.N * sp.nbk;
    const int chunk8 = (int)(gridDim.x >> 3);
    const int sw = (int)(blockIdx.x & 7) * chunk8 + (int)(blockIdx.x >> 3);
    if (sw >= per * sp.M) return;
    const int m = sw / per, rem = sw - m * per;
    const int b = rem / sp.nbk, k = rem - b * sp.nbk;
    const int bm = b * sp.M + m;
    const u32x2 rd = sp.red[(size_t)bm * sp.nbk + k];
    if (rd.y <= 1u) return;
    const int tid = threadIdx.x, row = tid >> 2, j4 = tid & 3;
    const unsigned pix = ((unsigned)k << kSortBPLog) + (unsigned)row;
    const float *src = sp.part + (((size_t)bm * sp.max_items + rd.x) * kSortBP + (unsigned)row) * 32u + (unsigned)j4 * 4u;
    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, c = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr size_t kStep = (size_t)kSortBP * 32u;
    unsigned sl = 0;
    for (; sl + 4 <= rd.y; sl += 4, src += 4 * kStep) {       // (four slices requested together, added in slice order)
        f32x4 x[4], y[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x[i] = *reinterpret_cast<const f32x4 *>(src + i * kStep);
            y[i] = *reinterpret_cast<const f32x4 *>(src + i * kStep + 16);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a += x[i];
            c += y[i];
        }
    }
    for (; sl < rd.y; ++sl, src += kStep) {
        a += *reinterpret_cast<const f32x4 *>(src);
        c += *reinterpret_cast<const f32x4 *>(src + 16);
    }
    if (pix >= (unsigned)sp.S) return;
    float *const dst = grad_value + ((((size_t)b * sp.S + pix) * sp.M + m) * 32u + (unsigned)j4 * 4u);
    *reinterpret_cast<f32x4 *>(dst) = a;
    *reinterpret_cast<f32x4 *>(dst + 16) = c;
}
