// msda_tile.h -- region tiling of self-attention over the pyramid: plan, tables and row helpers shared by the
// level-split backward (msda_bwd_tile_lv.h) and the counting-sort backward (msda_bwd_bins.h).
// Included by msda_hip.hip inside its anonymous namespace.
#pragma once

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
