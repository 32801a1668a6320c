// msda_hip.hip -- multi-scale deformable attention for gfx950 (MI355X, CDNA4).
//
// Hand-written HIP; wave64, LDS-staged sampling records, buffer (SRSRC) gathers with
// hardware zero padding, DPP reductions, fixed-point LDS accumulation, hardware f32 atomics.
// No CUDA-compat layer.
//
// Semantics replaced (reference repository paths):
//   forward   models/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299 (+ bilinear :33-84)
//   backward  models/ops/src/cuda/ms_deform_im2col_cuda.cuh:301-403 (+ bilinear :87-159)
//   host side models/ops/src/cuda/ms_deform_attn_cuda.cu:20-153
//   fused prologue (msda_fused_*): models/ops/modules/ms_deform_attn.py:104-123 -- softmax over the L*P
//             logits, sampling locations from offsets + reference points, padding-mask fill of `value`
// C ABI: include/msda_hip.h.  Design notes, byte counts and rooflines: DESIGN.md.
//
// Sources: this file holds the host side (options, kernel selection, launch planning, the C ABI); the kernels live
// in msda_generic.h (any D / dtype), msda_fwd_gather.h (D = 32 through the vector L1), msda_fwd_win.h (pyramid
// self-attention forward, LDS windows), msda_tile.h + msda_bwd_tile_lv.h / msda_bwd_bins.h (pyramid backward),
// msda_bwd_rows.h (decoder-shaped backward), msda_fused_side.h (Jacobian side kernels), msda_select.h (statistics).
//
// Variant numbers (msda_set_option "fwd_variant" / "bwd_variant"; 0 = auto):
//   forward : 0 auto (fp32 pyramid self-attention with host shapes: 12 unless "fwd_win_auto" is 0 or the selector asks
//             for the gather; other D = 32 calls: 3) | 1 generic | 2,3,4 d32 gather with 2,4,1 points in flight |
//             12 msda_fwd_d32_win (msda_fwd_win.h): (batch, head, region) per workgroup, windows of levels 1-3 filled
//             by LDS-DMA around the measured mean offset, level 0 through the vector L1, pixel-pair LDS reads, one
//             head per XCD
//   backward: 0 auto (pyramid self-attention: 12, or rows / 10 by selector level and call form; other D = 32 calls:
//             msda_bwd_d32_rows, 32 lanes per row) | 1 generic | 10 region-tiled fixed-point windows, one pyramid
//             level per workgroup | 12 counting sort + register gather (msda_bwd_bins.h) | 13 global sort + gather
//             through a workspace, no fabric atomics (msda_bwd_sorted.h; selector level 2 when the caller gave scratch)
//   (rounds 1-2 also had a hybrid LDS/L1 forward -- 8, 9 -- and an all-levels-per-workgroup backward -- 8, 9, 11:
//    measured slower than what replaced them, DESIGN.md 4.1 / 4.2, and removed in round 5; those numbers now mean 0)
//
// Kernel families
//   *_generic   any D/L/P, f32 / f64 / bf16 storage: one thread per output scalar
//               (forward) or one block per (n,q,m) row (backward).  Correctness path for
//               shapes the specialised kernels do not cover (reference gradcheck sizes
//               D in {30,64,71,1025,...}).
//   *_d32_win / *_d32_tile_* / *_d32_rows   region- or row-organised specialisations, described at their definitions
//               (msda_fwd_win.h, below, msda_bwd_rows.h)
//   *_d32       MeMOTR geometry (D = 32 channels/head): the lanes that own one (n,q,m) row hold its
//               32 channels (8 lanes x 4 fp32 channels, 4 lanes x 8 bf16 channels).  Each lane prepares
//               the sampling record of a share of the row's L*P points exactly once, parks it in LDS, and the
//               lanes of the row then stream the records back as broadcast ds_read_b128.  Corner reads
//               are 16-byte buffer loads; invalid corners carry an out-of-range offset so the buffer unit
//               returns zeros (= the reference's per-corner zero padding, no divergent branches).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <type_traits>
#include <vector>

#include "../../include/msda_hip.h"
#include "msda_common.h"
#include "msda_select.h"

namespace {

using namespace msda;

template <typename T>
__device__ __forceinline__ void atomic_add_hw(T *p, T v) {
    unsafeAtomicAdd(p, v);  // global_atomic_add_f32 / _f64, no CAS loop
}

__device__ __forceinline__ float t_exp(float x) { return expf(x); }
__device__ __forceinline__ double t_exp(double x) { return exp(x); }

#include "msda_generic.h"
#include "msda_fwd_gather.h"
#include "msda_tile.h"
#include "msda_bwd_tile_lv.h"
#include "msda_fused_side.h"
#include "msda_fwd_win.h"
#include "msda_bwd_rows.h"
#include "msda_bwd_bins.h"
#include "msda_bwd_sorted.h"

// ----------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------
thread_local char g_err[256] = {0};
thread_local const char *g_kernel = "";

std::atomic<int> opt_fwd_variant{0}, opt_bwd_variant{0};
std::atomic<int> opt_fwd_block{256}, opt_bwd_block{256};
std::atomic<int> opt_fwd_grid_mult{32}, opt_bwd_grid_mult{16};
// (backward margin 4, round 3: +1 % at the initialisation's offsets, -21 / -37 % when they are 1.5x / 2x larger,
//  profiles/r03_bwd_margin_sweep.txt)
std::atomic<int> opt_bwd_tile_margin{4};
std::atomic<int> opt_bwd_split{1};        // fused backward with a workspace: prologue kernel + plain tiled kernel + finish kernel
std::atomic<int> opt_bwd_wide_log2{12};   // tiled backward: row-magnitude range (log2) that makes a region "wide"; 0 = off
std::atomic<int> opt_bwd_ablate{0};
std::atomic<int> opt_bwd_rows_block{0};   // threads per workgroup of msda_bwd_d32_rows (0: by problem size)
std::atomic<int> opt_bwd_bins_strip{4};   // counting-sort backward: region rows per strip of the block -> region walk
std::atomic<int> opt_bwd_bins_margin{6};     // small-margin level (level 0 of the selector)
std::atomic<int> opt_bwd_bins_margin_hi{9};  // large-margin level (level 1): the largest window that keeps 5 workgroups per CU
std::atomic<int> opt_deterministic{0};       // 1: the forward keeps ONE summation order whatever the statistics say (pyramid: the windowed kernel)
std::atomic<int> opt_auto_select{1};         // msda_select.h: follow the measured off-window share (0: level 0 always)
std::atomic<int> opt_sel_level{-1};          // >= 0: pin the selector's level (tests, benchmarks)
// backward thresholds, 1/1000 of the valid corners.  Level 2 is the sorted backward when the caller provides scratch
// (0.21-0.28 ms at every offset scale, profiles/r06_sorted_probe.txt: it overtakes the large-margin windows once ~0.2 % of
// the corners leave them) and the rows kernel's float atomics otherwise (0.5-1.1 ms: only past 10 %)
std::atomic<int> opt_sel_up0{5}, opt_sel_up1{2}, opt_sel_down1{2}, opt_sel_down2{1};
std::atomic<int> opt_sel_up1_rows{100}, opt_sel_down2_rows{60};
std::atomic<int> opt_sel_fwd_up{50}, opt_sel_fwd_down{20};                                  // forward thresholds  // counting-sort backward: window margin (the window is only a table of counters)
std::atomic<int> opt_bwd_soft{1};         // fused counting-sort backward: softmax + its Jacobian in the kernel when the caller passes the forward's output
std::atomic<int> opt_bwd_sorted{1};       // selector level 2: grad_value by sort + gather when the caller gave scratch (0: the rows kernel)
std::atomic<int> opt_bwd_sort_qc{0}, opt_bwd_sort_emult{0};   // sorted backward: queries per dots workgroup / chunks per emit workgroup (0: auto)
std::atomic<int> opt_bwd_rows{1};         // 0: few-query D = 32 calls keep the generic row-per-block backward
std::atomic<int> opt_fwd_head_major{0};    // gather forward: head-major task walk (one head per XCD)
std::atomic<int> opt_fwd_win_rlog{0};       // windowed forward: log2 of the region height on level 0 (0: auto)
std::atomic<int> opt_fwd_win_rlogx{0};      // log2 of the region width (at least the height; 0: as the height)
std::atomic<int> opt_fwd_win_bf16{0};       // 1: bf16 rows take the windowed forward too (built and measured in round 6: 51.0 vs 49.2 us
                                            // for the gather kernel, fp32 windows 45.6 -- profiles/r06_bf16_fwd_probe.txt; default: the gather)
std::atomic<int> opt_fwd_win_auto{1};       // 0: never pick the windowed forward on its own
std::atomic<int> opt_fwd_win_block{0};      // threads per workgroup (128 / 256 / 384 / 512; 0: auto)
std::atomic<int> opt_fwd_win_l0{1};         // first level served from an LDS window
std::atomic<int> opt_fwd_win_margins{0x3333};  // window margin per level, 4 bits each (level 0 in the low nibble)
std::atomic<int> opt_fwd_win_ablate{0};     // profiling only
std::atomic<int> opt_fwd_win_place{0};      // 1: measured window placement in every workgroup (rounds 3-4)
std::atomic<int> opt_fwd_win_grid{1};       // 1: the default shape may become equal regions of any size when that fills the slots in fewer rounds (0: never)
std::atomic<int> opt_fwd_win_rsy{0}, opt_fwd_win_rsx{0};   // region height / width on level 0 in pixels: grid mode with exactly this size (0: by estimate)
std::atomic<int> opt_fwd_win_wps{0};        // 3 / 4: force the 168- / 128-register build; 2 (256 threads): the 256-register build
std::atomic<int> opt_fwd_win_early{9};      // 0 / 2 / 4: level-0 points requested before the LDS phase (else: by register budget)
constexpr int kWinEarlyW4 = 0;              // ... of the 128-register build (0 and 2 time the same; 0 needs 107 registers, no spill)
std::atomic<int> opt_bwd_side_rows{1};      // slim split backward: side kernels with one lane per (query, head) row (0: one lane per point)
std::atomic<int> opt_fwd_win_trace_lo{0}, opt_fwd_win_trace_hi{0};   // profiling: device address of the timeline buffer (31 + 31 bits)
       // profiling only: drop parts of the tiled backward (results are then wrong)

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

// the specialised kernels address `value` (and the fp32 grad_value) with 32-bit byte offsets
bool d32_ok(int D, int L, long value_elems) { return D == 32 && L <= kMaxLevels && value_elems * 4 < 0x7fffff00L; }

// Plan the region tiling from the HOST copy of the level shapes.  Returns false when the tiled
// kernels do not apply (then the gather / generic kernels run).  Levels below `l0` get no window.
bool make_tile_plan(TilePlan &pl, const int64_t *shapes_host, int N, int S, int M, int D, int L, int Lq, int P,
                    long value_bytes, int margin, int l0, size_t lds_rows_extra, size_t fixed_lds, size_t &lds,
                    bool check_lds = true) {
    if (!shapes_host || D != 32 || L < 1 || L > kTileMaxL || Lq != S || L * P > kMaxFusedLP) return false;
    if (margin < 0) margin = 0;
    if (l0 < 0) l0 = 0;
    if (l0 > L) l0 = L;
    memset(&pl, 0, sizeof(pl));
    pl.N = N; pl.S = S; pl.M = M; pl.L = L; pl.P = P; pl.Lq = Lq; pl.l0 = l0;
    pl.value_bytes = (unsigned)value_bytes;
    long q = 0;
    int rows = 0, px = 0, RY = 0, RX = 0;
    for (int l = 0; l < kTileMaxL; ++l) {
        if (l < L) {
            const long H = shapes_host[2 * l], W = shapes_host[2 * l + 1];
            if (H <= 0 || W <= 0 || H > 32767 || W > 32767 || W * M >= (1L << 23)) return false;
            const int sh = L - 1 - l, side = 1 << sh;
            int win = side + 2 * margin;
            if (win > 32) win = 32;
            if (l < l0) win = 0;
            pl.H[l] = (int)H; pl.W[l] = (int)W; pl.qstart[l] = (int)q; pl.shift[l] = sh; pl.row0[l] = rows;
            pl.win[l] = win; pl.win_base[l] = px;
            int magic = 65537;
            if (win > 0) {
                magic = 65536 / win + 1;
                for (int x = 0; x < win * win; ++x)
                    if (((x * magic) >> 16) != x / win) return false;
            }
            pl.win_magic[l] = magic;
            q += H * W; rows += side * side; px += win * win;
            const int ry = (int)((H + side - 1) / side), rx = (int)((W + side - 1) / side);
            RY = ry > RY ? ry : RY;
            RX = rx > RX ? rx : RX;
        } else {  // inert padding so that table loads stay in range
            pl.H[l] = 1; pl.W[l] = 1; pl.qstart[l] = (int)q; pl.shift[l] = 0; pl.row0[l] = rows;
            pl.win[l] = 0; pl.win_magic[l] = 65537; pl.win_base[l] = px;
        }
    }
    if (q != S) return false;  // host shapes do not describe this value tensor
    if (px + (int)lds_rows_extra >= 65535) return false;    // window cells travel as 16-bit indices
    pl.row0[kTileMaxL] = rows; pl.win_base[kTileMaxL] = px;
    for (int l = L; l <= kTileMaxL; ++l) { pl.row0[l] = rows; pl.win_base[l] = px; }
    pl.rows = rows; pl.RY = RY; pl.RX = RX;
    if (rows > kTileMaxRows) return false;
    const long nb = (long)N * RY * RX * M;
    if (nb > (1L << 30)) return false;
    pl.n_blocks = (int)nb;
    lds = ((size_t)px + lds_rows_extra) * 128 + fixed_lds;
    return !check_lds || lds <= 160 * 1024 - 2048;
}

// LDS layout of msda_bwd_d32_bins (msda_bwd_bins.h) for `ni` items per thread; false when the call does not fit.
bool make_bins_plan(BinsPlan &bp, const TilePlan &pl, int ni, size_t &lds) {
    const int P = pl.P, n_items = pl.rows * P;
    if (P < 1 || n_items < 1 || n_items > kTileThreads * ni) return false;
    const int magic = 65536 / P + 1;
    for (int i = 0; i < kTileThreads * ni; ++i)
        if (((i * magic) >> 16) != i / P) return false;
    int win_max = 0;
    for (int l = 0; l < pl.L; ++l) win_max = pl.win[l] > win_max ? pl.win[l] : win_max;
    const int ncell = win_max * win_max;
    const int C = (((ncell + 63) / 64) + 3) & ~3;
    auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    size_t o = up16((size_t)(pl.rows + 1) * kBinsGRow);
    bp.n_items = n_items;
    bp.magic_p = magic;
    bp.scan_c = C;
    bp.strip = opt_bwd_bins_strip.load() > 0 ? opt_bwd_bins_strip.load() : 1;
    bp.o_x = (unsigned)o;                       // records + flags, later the sorted entries
    bp.o_fl = (unsigned)(o + (size_t)n_items * 16);
    const size_t rec_bytes = up16((size_t)n_items * 20), ent_bytes = up16((size_t)(4 * n_items + 1) * 8);
    o += rec_bytes > ent_bytes ? rec_bytes : ent_bytes;
    // union: ticket counters + row tables (dead after the sort) | flush transpose
    bp.o_st = (unsigned)o;
    bp.o_cnt = (unsigned)o;
    bp.o_rowp = (unsigned)(o + (size_t)C * 64 * 4);
    bp.o_rowa = (unsigned)(bp.o_rowp + up16((size_t)(pl.rows + 1) * 4));
    bp.o_rowq = (unsigned)(bp.o_rowa + up16((size_t)(pl.rows + 1) * 4));
    const size_t tab_bytes = (size_t)C * 64 * 4 + 3 * up16((size_t)(pl.rows + 1) * 4);
    const size_t st_bytes = (size_t)(kTileThreads / 64) * kBinsStageWave;
    o += tab_bytes > st_bytes ? tab_bytes : st_bytes;
    bp.o_start = (unsigned)o;
    o += up16((size_t)C * 64 * 2);
    bp.o_comp = (unsigned)o;
    o += (size_t)C * 64 * 4;
    bp.o_misc = (unsigned)o;
    o += 16 + (size_t)(kTileThreads / 64) * 16 * 4 + 16;
    o = up16(o);
    bp.o_dot = (unsigned)o;                     // (soft) <grad_out_row, out_row> per staged row
    o += up16((size_t)(pl.rows + 1) * 4);
    lds = o;
    return lds <= 64 * 1024;
}

// ---- grid mode of the windowed forward: which region size, if any ----
// Estimate of a launch in "row units": a workgroup costs kWinPrologueRows + its rows, the workgroups of one XCD (an eighth
// of the launch: head-major numbering, one contiguous run per XCD) are handed to its 64 slots (32 CUs x two 512-thread
// workgroups) in dispatch order as slots free up.  The power-of-two plan's partial border regions come last; equal
// regions all cost the same.  Grid mode is taken when its best size is at least 7 % faster by this estimate AND keeps two
// workgroups per CU (LDS); the choice is cached per geometry.
constexpr int kWinPrologueRows = 140;      // (prologue ~ 10 of the 34 us a 340-row workgroup takes under load)

inline double win_makespan(const std::vector<int> &dur_one, long repeats, int slots) {
    std::vector<double> free_at((size_t)slots, 0.0);      // a binary heap by hand would be faster; this runs once per geometry
    double end = 0.0;
    for (long r = 0; r < repeats; ++r)
        for (int d : dur_one) {
            size_t k = 0;
            for (size_t i = 1; i < free_at.size(); ++i)
                if (free_at[i] < free_at[k]) k = i;
            free_at[k] += (double)(kWinPrologueRows + d);
            end = free_at[k] > end ? free_at[k] : end;
        }
    return end;
}

struct WinGridChoice {
    long key[12];
    int rsy, rsx;
};

inline void win_grid_choice(const WinPlan &p2, const int64_t *shapes_host, int N, int S, int M, int D, int L, int Lq, int P,
                            long value_bytes, int lwin0, const int *margins, int threads, int margin_bits, int &rsy,
                            int &rsx) {
    rsy = rsx = 0;
    if (L < 1 || L > kWinMaxL || (M % 8) != 0) return;
    static std::mutex mu;
    static std::vector<WinGridChoice> cache;
    WinGridChoice c;
    memset(&c, 0, sizeof(c));
    for (int l = 0; l < L; ++l) { c.key[2 * l] = shapes_host[2 * l]; c.key[2 * l + 1] = shapes_host[2 * l + 1]; }
    c.key[8] = N; c.key[9] = M; c.key[10] = ((long)L << 40) | ((long)P << 32) | ((long)lwin0 << 24) | (long)margin_bits;
    c.key[11] = threads;
    {
        std::lock_guard<std::mutex> g(mu);
        for (const WinGridChoice &e : cache)
            if (!memcmp(e.key, c.key, sizeof(c.key))) { rsy = e.rsy; rsx = e.rsx; return; }
    }
    const int slots = 64;
    const long per_xcd = (long)N * M / 8;              // (head, image) pairs per XCD
    // the power-of-two plan in its dispatch order: complete regions, right border column, bottom border row
    std::vector<int> d2;
    {
        auto rows_of = [&](int ry, int rx) {
            int rows = 0;
            for (int l = 0; l < L; ++l) {
                const int sy = p2.shy[l], sx = p2.shx[l];
                int hv = p2.H[l] - (ry << sy), wv = p2.W[l] - (rx << sx);
                hv = hv > (1 << sy) ? (1 << sy) : (hv < 0 ? 0 : hv);
                wv = wv > (1 << sx) ? (1 << sx) : (wv < 0 ? 0 : wv);
                rows += hv * wv;
            }
            return rows;
        };
        for (int ry = 0; ry < p2.RYf; ++ry) for (int rx = 0; rx < p2.RXf; ++rx) d2.push_back(rows_of(ry, rx));
        for (int ry = 0; ry < p2.RYf; ++ry) for (int rx = p2.RXf; rx < p2.RX; ++rx) d2.push_back(rows_of(ry, rx));
        for (int ry = p2.RYf; ry < p2.RY; ++ry) for (int rx = 0; rx < p2.RX; ++rx) d2.push_back(rows_of(ry, rx));
    }
    const double cost2 = win_makespan(d2, per_xcd, slots);
    double best = cost2 * 0.93;
    const long H0 = shapes_host[0], W0 = shapes_host[1];
    for (int ty = 4; ty <= 40; ++ty)
        for (int tx = 4; tx <= 32; ++tx) {
            const long nreg = ((H0 + ty - 1) / ty) * ((W0 + tx - 1) / tx);
            if (nreg * per_xcd < slots / 2) continue;                   // too few workgroups to fill the XCD at all
            // a lower bound before the plan: all regions at most ty tx (1 + 1/4 + 1/16 + 1/64) rows
            WinPlan g;
            size_t glds = 0;
            if (!make_win_plan_grid(g, shapes_host, N, S, M, D, L, Lq, P, value_bytes, ty, tx, lwin0, margins, threads, glds,
                                    4))
                continue;
            if (glds + 512 > 80 * 1024) continue;                       // two workgroups per CU, as the default shape
            const long w = nreg * per_xcd;
            const double cost = (double)((w + slots - 1) / slots) * (double)(kWinPrologueRows + g.rows);
            if (cost < best) { best = cost; rsy = ty; rsx = tx; }
        }
    c.rsy = rsy; c.rsx = rsx;
    std::lock_guard<std::mutex> g(mu);
    if (cache.size() >= 64) cache.erase(cache.begin());
    cache.push_back(c);
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
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    done.fetch_or(bit, std::memory_order_release);
    return MSDA_OK;
}

// ---- msda_select.h, host side: the records and their table ----
SelSlot g_sel[kSelSlots];
std::mutex g_sel_mu;
unsigned long long *g_sel_pool_dev[64] = {nullptr};     // per device ordinal: kSelSlots records, one allocation
unsigned long long *g_sel_pool_host[64] = {nullptr}, *g_sel_pool_host_dev[64] = {nullptr};
bool g_sel_pool_failed[64] = {false};
unsigned long long g_sel_clock = 0, g_sel_tick = 0;
thread_local unsigned long long g_site = 0;
// msda_next_value_pixel_stride: elements between two pixels of `value` (and of `grad_value`) for the NEXT forward /
// backward call of this thread (0: contiguous, M * D).  Read once -- by that call, whatever its outcome -- and cleared.
thread_local long g_value_stride = 0;
inline long take_value_stride() {
    const long v = g_value_stride;
    g_value_stride = 0;
    return v;
}
thread_local int g_sel_level = 0;           // level of this thread's last selected call (msda_selector_last)
thread_local float g_sel_frac = -1.f, g_sel_inner = -1.f;

bool stream_capturing(hipStream_t stream) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess) {
        (void)hipGetLastError();
        return true;
    }
    return cs != hipStreamCaptureStatusNone;
}

// The device's block of records: ONE hipMalloc + one hipHostMalloc per device for the life of the process, made at the
// first call that is not inside a stream capture (allocation is not capturable).  g_sel_mu held.
bool sel_pool(int dev, hipStream_t stream) {
    const int d = dev & 63;
    if (g_sel_pool_dev[d]) return true;
    if (g_sel_pool_failed[d] || stream_capturing(stream)) return false;
    unsigned long long *pd = nullptr, *ph = nullptr, *phd = nullptr;
    const size_t dbytes = (size_t)kSelSlots * kSelDevWords * 8, hbytes = (size_t)kSelSlots * kSelHostWords * 8;
    if (hipMalloc((void **)&pd, dbytes) != hipSuccess || hipHostMalloc((void **)&ph, hbytes, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&phd, ph, 0) != hipSuccess || hipMemset(pd, 0, dbytes) != hipSuccess) {
        (void)hipGetLastError();
        if (pd) (void)hipFree(pd);
        if (ph) (void)hipHostFree(ph);
        g_sel_pool_failed[d] = true;
        return false;
    }
    memset(ph, 0, hbytes);
    g_sel_pool_dev[d] = pd; g_sel_pool_host[d] = ph; g_sel_pool_host_dev[d] = phd;
    return true;
}

// The record of a call site; null when the mechanism is off or the device's block does not exist yet (first call of
// the process inside a capture): the call then runs at level 0 without statistics.  A full table hands the least
// recently used record to the new key (its counters keep counting: the first record read is only a baseline; the
// forward's window placement words are cleared).
SelSlot *sel_acquire(int kind, int M, int L, int P, int dt, hipStream_t stream) {
    if (!opt_auto_select.load()) return nullptr;
    SelKey k;
    memset(&k, 0, sizeof(k));
    (void)hipGetDevice(&k.dev);
    k.kind = kind; k.site = g_site; k.M = M; k.L = L; k.P = P; k.dt = dt;
    std::lock_guard<std::mutex> lock(g_sel_mu);
    SelSlot *pick = nullptr;
    for (int i = 0; i < kSelSlots; ++i) {
        if (g_sel[i].used && g_sel[i].key == k) {
            g_sel[i].stamp = ++g_sel_clock;
            return &g_sel[i];
        }
        if (!g_sel[i].used) {
            if (!pick || pick->used) pick = &g_sel[i];
        } else if (g_sel[i].key.dev == k.dev && (!pick || (pick->used && g_sel[i].stamp < pick->stamp))) {
            pick = &g_sel[i];
        }
    }
    if (!pick || !sel_pool(k.dev, stream)) return nullptr;
    const int d = k.dev & 63, i = (int)(pick - g_sel);
    SelSlot &s = *pick;
    if (s.used) {
        // A record that changes hands keeps counting (the first read is a baseline), but the window placement the
        // old call site measured -- mean offsets and their "measured" bits -- would centre the new site's windows on
        // another module's offsets and report what leaves them: those words start over.  A memset cannot be recorded
        // into a capture (it would replay); a capturing call with no record of its own runs without one.
        if (stream_capturing(stream)) return nullptr;
#ifndef MSDA_SEL_KEEP_PLACEMENT      // (test builds only: the bug the reuse test pins)
        unsigned long long *const rec = g_sel_pool_dev[d] + (size_t)i * kSelDevWords;
        if (hipMemsetAsync(rec + kSelHintAccWord, 0, (size_t)(kSelHintValidWord + 1 - kSelHintAccWord) * 8, stream) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
#endif
    }
    s.key = k;
    s.dev = g_sel_pool_dev[d] + (size_t)i * kSelDevWords;
    s.host = g_sel_pool_host[d] + (size_t)i * kSelHostWords;
    s.host_dev = g_sel_pool_host_dev[d] + (size_t)i * kSelHostWords;
    s.primed = false;
    s.seen = s.host[9];
    memset(s.last, 0, sizeof(s.last));
    s.level = s.eff = 0; s.calls = 0u; s.frac = s.frac_inner = -1.f;      // (-1: nothing measured yet)
    s.polled = false; s.pub_seen = s.seen; s.scratch = false;
    s.stamp = ++g_sel_clock;
    s.used = true;
    return &s;
}

SelRule sel_rule(int kind, bool scratch = true) {
    SelRule r;
    if (kind == 0) { r.up0 = opt_sel_fwd_up.load(); r.down1 = opt_sel_fwd_down.load(); r.up1 = r.down2 = 0; }
    else {
        r.up0 = opt_sel_up0.load(); r.down1 = opt_sel_down1.load();
        r.up1 = scratch ? opt_sel_up1.load() : opt_sel_up1_rows.load();
        r.down2 = scratch ? opt_sel_down2.load() : opt_sel_down2_rows.load();
    }
    return r;
}

// Read what the launches have left in the host record and move the level (g_sel_mu held).  Per level the launches ran
// at, the difference to the counters seen last is judged once it holds kSelMinSample valid corners: a probe ran one
// level below the top and is judged by the top level's rule; counts of a level the record has moved away from are
// dropped.
void sel_refresh(SelSlot *s) {
    const unsigned long long seq = s->host[9];
    if (seq == s->seen) return;
    std::atomic_thread_fence(std::memory_order_acquire);
    unsigned long long cur[kSelLevels][3];
    for (int l = 0; l < kSelLevels; ++l)
        for (int c = 0; c < 3; ++c) cur[l][c] = s->host[l * 3 + c];
    s->seen = seq;
    if (!s->primed) {
        memcpy(s->last, cur, sizeof(cur));
        s->primed = true;
        return;
    }
    const int kind = s->key.kind, top = kind == 0 ? 1 : 2;
    const SelRule r = sel_rule(kind, s->scratch);
    for (int ran = 0; ran < kSelLevels; ++ran) {
        const unsigned long long dv = cur[ran][0] - s->last[ran][0];
        if (dv < kSelMinSample || dv > (1ull << 62)) continue;
        const float f = (float)((double)(cur[ran][1] - s->last[ran][1]) / (double)dv);
        const float fi = (float)((double)(cur[ran][2] - s->last[ran][2]) / (double)dv);
        memcpy(s->last[ran], cur[ran], sizeof(cur[ran]));
        if (s->level == top && ran == top - 1) {
            s->frac = f; s->frac_inner = fi;
            s->level = sel_next_level(kind, top, f * 1000.f, fi * 1000.f, r);
        } else if (ran == s->level) {
            s->frac = f; s->frac_inner = fi;
            s->level = sel_next_level(kind, ran, f * 1000.f, fi * 1000.f, r);
        }
    }
}

// The level for THIS call.  Eager calls count themselves: a level without windows probes one level down every
// kSelProbeEvery-th call (`probe` is then set).  A capturing call takes what the last msda_selector_poll() announced.
int sel_level(SelSlot *s, int kind, bool &probe, bool capturing = false) {
    probe = false;
    const int pinned = opt_sel_level.load();
    if (!s) return pinned >= 0 ? pinned : 0;
    std::lock_guard<std::mutex> lock(g_sel_mu);
    sel_refresh(s);
    g_sel_frac = s->frac;
    g_sel_inner = s->frac_inner;
    const int top = kind == 0 ? 1 : 2;
    int level;
    if (capturing) {
        level = pinned >= 0 ? pinned : s->eff;
        probe = pinned < 0 && level != s->level;
    } else {
        ++s->calls;
        level = pinned >= 0 ? pinned : s->level;
        // (`eff` belongs to msda_selector_poll() once a caller polls: the eager warm-up calls a graph cache makes in
        //  front of a capture used to reset it here, and the capture keyed on "probe one level down" then held the
        //  top level's kernel -- advisor, round 5.  Without a polling caller a capture takes the level of the last
        //  eager call, as before)
        if (!s->polled) s->eff = s->level;
        if (pinned < 0 && level == top && (s->calls % kSelProbeEvery) == 0u) {
            level = top - 1;
            probe = true;
        }
    }
    g_sel_level = level;
    return level;
}

// What sel_level() would answer for this thread's call site, without counting a call or creating a record (the
// workspace query: a caller sizes its scratch before the call).
int sel_peek(int kind, int M, int L, int P, int dt, hipStream_t stream) {
    const int pinned = opt_sel_level.load();
    if (pinned >= 0) return pinned;
    if (!opt_auto_select.load()) return 0;
    SelKey k;
    memset(&k, 0, sizeof(k));
    (void)hipGetDevice(&k.dev);
    k.kind = kind; k.site = g_site; k.M = M; k.L = L; k.P = P; k.dt = dt;
    const bool capturing = stream_capturing(stream);
    std::lock_guard<std::mutex> lock(g_sel_mu);
    for (int i = 0; i < kSelSlots; ++i)
        if (g_sel[i].used && g_sel[i].key == k) {
            g_sel[i].scratch = true;        // (a caller that asks will bring scratch: level 2 is then the sorted backward)
            sel_refresh(&g_sel[i]);
            return capturing ? g_sel[i].eff : g_sel[i].level;
        }
    return 0;
}

struct FusedArgs {     // null proj = the plain operator
    const float *proj = nullptr;
    int proj_stride = 0;
    const float *ref = nullptr;
    int ref_dim = 0;
    const unsigned char *mask = nullptr;
};

PointSrc make_src(const void *loc, const void *attn, const FusedArgs &fa, int M, int L, int P) {
    PointSrc s;
    s.loc = (const float *)loc;
    s.attn = (const float *)attn;
    s.proj = fa.proj;
    s.ref = fa.ref;
    s.mask = fa.mask;
    s.proj_stride = fa.proj_stride;
    s.n_off = 2 * M * L * P;
    s.ref_dim = fa.ref_dim;
    return s;
}

int check_fused(const FusedArgs &fa, int M, int L, int P) {
    if (!fa.proj || !fa.ref) return fail(MSDA_EINVAL, "null pointer argument");
    if (fa.ref_dim != 2 && fa.ref_dim != 4) return fail(MSDA_EINVAL, "reference points must have 2 or 4 coordinates");
    if (fa.proj_stride < 3 * M * L * P) return fail(MSDA_EINVAL, "projection rows shorter than 3*M*L*P");
    if (L * P > kMaxFusedLP) return fail(MSDA_ENOTSUP, "fused prologue supports at most 64 sampling points per head");
    return MSDA_OK;
}

template <typename TV, typename TC>
int forward_impl(const TV *value, const int64_t *shapes, const int64_t *lstart, const TC *loc, const TC *attn,
                 const FusedArgs &fa, int N, int S, int M, int D, int L, int Lq, int P, TV *out,
                 const int64_t *shapes_host, hipStream_t stream) {
    const long vstride = take_value_stride();
    const bool fused = fa.proj != nullptr;
    int rc = fused ? check_dims(value, shapes, lstart, fa.proj, fa.ref, out, N, S, M, D, L, Lq, P)
                   : check_dims(value, shapes, lstart, loc, attn, out, N, S, M, D, L, Lq, P);
    if (rc) return rc;
    if (fused && (rc = check_fused(fa, M, L, P))) return rc;
    if ((long)N * Lq == 0) { g_err[0] = 0; return MSDA_OK; }
    int variant = opt_fwd_variant.load();
    constexpr bool kD32Type = sizeof(TC) == 4 && (sizeof(TV) == 4 || sizeof(TV) == 2);
    // `value` as a slice of a wider tensor (msda_next_value_pixel_stride): the gather kernel only -- the call sites that
    // use it are the decoder's (a few hundred queries), which take that kernel anyway
    const bool strided = vstride != 0 && vstride != (long)M * D;
    const long value_elems = strided ? (long)N * S * vstride : (long)N * S * M * D;
    const long value_bytes = value_elems * (long)sizeof(TV);
    const bool can32 = kD32Type && d32_ok(D, L, value_elems);
    if (strided) {
        if (vstride < (long)M * D || (vstride * (long)sizeof(TV)) % 16 != 0)
            return fail(MSDA_EINVAL, "value pixel stride: at least M * D elements, rows 16-byte aligned");
        if (!can32) return fail(MSDA_ENOTSUP, "value pixel stride: D = 32 float32 / bfloat16 calls only");
        variant = 3;
    }
    SelSlot *slot = nullptr;
    int sel = 0;
    bool sel_head_major = false;
    if (variant == 0) {
        // self-attention over the pyramid (one query per pixel): coarse levels from per-head LDS windows; every
        // other D = 32 call: direct gather with 4 points (16 rows) in flight -- best of the sweeps in profiles/
        // (bf16 rows, round 6: the same kernel on 64-byte rows exists -- "fwd_win_bf16" 1 or "fwd_variant" 12 -- and is
        //  slower than the gather kernel: halving the LDS bytes bought nothing, the widening costs VALU)
        const bool pyramid = can32 && (sizeof(TV) == 4 || opt_fwd_win_bf16.load() != 0) && shapes_host != nullptr && Lq == S &&
                             L <= kWinMaxL && L * P <= 16 && opt_fwd_win_auto.load() != 0;
        variant = pyramid ? 12 : (can32 ? 3 : 1);
        if (pyramid) {      // msda_select.h: windows while the points stay near their queries, else the head-major gather
            slot = sel_acquire(0, M, L, P, (int)sizeof(TV), stream);
            bool probe = false;
            sel = sel_level(slot, 0, probe, slot != nullptr && stream_capturing(stream));
            // "deterministic": the windowed and the gather kernel add a row's points in different orders (same values to
            // 2e-5, different last bits), and the selector moves a call site between them from statistics of EARLIER calls;
            // with the option on, identical calls return identical bits -- the windowed kernel's, whose results do not depend
            // on where its windows sit (the record keeps measuring, nothing follows it)
            if (opt_deterministic.load()) sel = 0;
            if (sel >= 1) { variant = 3; sel_head_major = true; }
        }
    } else if (variant == 12 && can32 && shapes_host != nullptr) {
        slot = sel_acquire(0, M, L, P, (int)sizeof(TV), stream);     // forced: the selector only measures
        bool probe = false;
        (void)sel_level(slot, 0, probe, slot != nullptr && stream_capturing(stream));
        sel = 0;
    }
    if (variant >= 2 && !can32) variant = 1;
    if (variant == 1) {
        const long total = (long)N * Lq * M * D;
        const int grid = clamp_grid((total + 255) / 256, 32);
        if constexpr (sizeof(TC) == 4) {
            const PointSrc src = make_src(loc, attn, fa, M, L, P);
            if (fused) {
                g_kernel = "msda_fwd_generic<fused>";
                hipLaunchKernelGGL((msda_fwd_generic<TV, TC, true>), dim3(grid), dim3(256), 0, stream, value, shapes,
                                   lstart, (const TC *)nullptr, (const TC *)nullptr, src, N, S, M, D, L, Lq, P, out);
                return check_launch(g_kernel);
            }
        }
        g_kernel = "msda_fwd_generic";
        hipLaunchKernelGGL((msda_fwd_generic<TV, TC, false>), dim3(grid), dim3(256), 0, stream, value, shapes, lstart,
                           loc, attn, PointSrc{}, N, S, M, D, L, Lq, P, out);
        return check_launch(g_kernel);
    }
    if constexpr (kD32Type) {
        const PointSrc src = make_src(loc, attn, fa, M, L, P);
        {
            if (variant == 12) {
                WinPlan wp;
                size_t lds = 0;
                const int mgs = opt_fwd_win_margins.load();
                const int margins[kWinMaxL] = {mgs & 15, (mgs >> 4) & 15, (mgs >> 8) & 15, (mgs >> 12) & 15};
                // Region shape and workgroup size (0 = auto).  Round 5: 16 x 16-pixel regions and 512 threads -- 340 rows
                // share one set of windows and one prologue (8 x 8: 85), two workgroups = 16 wavefronts per CU at 128
                // registers; border regions hold only the rows that exist and are walked last, so the 616 workgroups
                // of one 800 x 1333 image end together on the 512 slots (46.0 vs 55.1 us fused, N = 5: 224 vs 289 us;
                // profiles/r05_fwd_win_sweep_*.txt).  A geometry that shape cannot take falls back to 8 x 8 / 256.
                int threads = opt_fwd_win_block.load();
                int rlogy = opt_fwd_win_rlog.load(), rlogx = opt_fwd_win_rlogx.load();
                const bool auto_shape = threads == 0 && rlogy == 0 && rlogx == 0;
                if (threads != 512 && threads != 384 && threads != 128 && threads != 256) threads = auto_shape ? 512 : 256;
                if (rlogy == 0) rlogy = auto_shape ? 4 : 3;
                if (rlogx == 0) rlogx = rlogy;
                constexpr int kMaskGroups = sizeof(TV) == 4 ? 8 : 4;      // fill groups a wavefront's lanes cover per level
                bool planned = make_win_plan(wp, shapes_host, N, S, M, D, L, Lq, P, value_bytes, rlogx, rlogy,
                                             opt_fwd_win_l0.load(), margins, threads, lds, (int)sizeof(TV)) &&
                               !(src.mask != nullptr && wp.wgroups_max > kMaskGroups * (threads / 64));
                // Equal regions of any size (make_win_plan_grid) where they fill the workgroup slots in fewer rounds than
                // the power-of-two ones: asked for by size ("fwd_win_rsy" / "fwd_win_rsx"), or chosen by estimate for the
                // default shape (win_grid_choice, cached per geometry)
                {
                    int rsy = opt_fwd_win_rsy.load(), rsx = opt_fwd_win_rsx.load();
                    if (rsy <= 0 && rsx <= 0 && auto_shape && planned && opt_fwd_win_grid.load() != 0 && sizeof(TV) == 4)
                        win_grid_choice(wp, shapes_host, N, S, M, D, L, Lq, P, value_bytes, opt_fwd_win_l0.load(), margins,
                                        threads, mgs, rsy, rsx);
                    if (rsy > 0 || rsx > 0) {
                        if (rsy <= 0) rsy = rsx;
                        if (rsx <= 0) rsx = rsy;
                        WinPlan gp_;
                        size_t glds = 0;
                        if (make_win_plan_grid(gp_, shapes_host, N, S, M, D, L, Lq, P, value_bytes, rsy, rsx,
                                               opt_fwd_win_l0.load(), margins, threads, glds, (int)sizeof(TV)) &&
                            !(src.mask != nullptr && gp_.wgroups_max > kMaskGroups * (threads / 64))) {
                            wp = gp_;
                            lds = glds;
                            planned = true;
                        }
                    }
                }
                if (!planned && auto_shape) {
                    threads = 256;
                    planned = make_win_plan(wp, shapes_host, N, S, M, D, L, Lq, P, value_bytes, 3, 3,
                                            opt_fwd_win_l0.load(), margins, threads, lds, (int)sizeof(TV)) &&
                              !(src.mask != nullptr && wp.wgroups_max > kMaskGroups * (threads / 64));
                }
                // (bf16 rows: the default shape only -- 512 threads, no early loads, no profiling build; else the gather)
                if (sizeof(TV) == 2 && (threads != 512 || opt_fwd_win_ablate.load() != 0 || opt_fwd_win_trace_lo.load() != 0 ||
                                        opt_fwd_win_trace_hi.load() != 0))
                    planned = false;
                if (planned) {
                    const int grid = (wp.n_blocks + 7) & ~7;
#define MSDA_LAUNCH_WIN(FU, WPS, NE, NAME)                                                                           \
    do {                                                                                                             \
        rc = allow_big_lds(msda_fwd_d32_win<TV, FU, WPS, NE>, lds);                                                  \
        if (rc) return rc;                                                                                           \
        g_kernel = NAME;                                                                                             \
        hipLaunchKernelGGL((msda_fwd_d32_win<TV, FU, WPS, NE>), dim3(grid), dim3(threads), lds, stream,              \
                           value, lstart, src, out, wp);                                                             \
    } while (0)
#define MSDA_LAUNCH_WIN_T(FU, WPS, NE, NAME)                                                                         \
    do {                                                                                                             \
        rc = allow_big_lds(msda_fwd_d32_win<TV, FU, WPS, NE, true>, lds);                                            \
        if (rc) return rc;                                                                                           \
        g_kernel = NAME;                                                                                             \
        hipLaunchKernelGGL((msda_fwd_d32_win<TV, FU, WPS, NE, true>), dim3(grid), dim3(threads), lds, stream,        \
                           value, lstart, src, out, wp);                                                             \
    } while (0)
#define MSDA_LAUNCH_WIN_P(FU, WPS, NE, PB, NAME)                                                                     \
    do {                                                                                                             \
        rc = allow_big_lds(msda_fwd_d32_win<TV, FU, WPS, NE, false, PB>, lds);                                       \
        if (rc) return rc;                                                                                           \
        g_kernel = NAME;                                                                                             \
        hipLaunchKernelGGL((msda_fwd_d32_win<TV, FU, WPS, NE, false, PB>), dim3(grid), dim3(threads), lds, stream,   \
                           value, lstart, src, out, wp);                                                             \
    } while (0)
                    wp.ablate = opt_fwd_win_ablate.load();
                    wp.trace = reinterpret_cast<unsigned long long *>(((unsigned long long)opt_fwd_win_trace_hi.load() << 31) |
                                                                      (unsigned long long)opt_fwd_win_trace_lo.load());
                    // The kernel's publisher is wavefront 1: a 64-thread workgroup (options only) runs without statistics.
                    // (The share's denominator counts a wavefront's staged rows sixteen steps per ballot, any step count.)
                    const int win_nw = threads / 64;
                    const bool stats_ok = slot != nullptr && win_nw >= 2;
                    if (stats_ok) {     // (cumulative counters, fixed addresses: a captured launch counts like an eager one)
                        wp.stats = slot->dev;
                        wp.stats_host = slot->host_dev;
                        wp.sel_level = sel;
                    }
                    // windows placed from the record's running mean offsets (no round trip in front of the fill); without
                    // a record, or on request, every workgroup measures its own first ("fwd_win_place" 1)
                    wp.measure = (!stats_ok || M > kSelHintHeads || L > kSelHintLevels || opt_fwd_win_place.load() != 0) ? 1 : 0;
                    // register budget by what the workgroup shape admits: three 256-thread workgroups per CU (40-53 KB
                    // of LDS each) -> 168 registers, all four level-0 points requested before the LDS phase; 512-thread
                    // workgroups (two per CU) or four small ones -> 128 registers, two of them
                    int wps = opt_fwd_win_wps.load();
                    const bool wide = wps == 2 && threads <= 256 && sizeof(TV) == 4 && !wp.trace && !wp.ablate;
                    if (wps != 3 && wps != 4) wps = (threads <= 256 && lds + 640 > 40 * 1024) ? 3 : 4;
                    int early = opt_fwd_win_early.load();          // 0 / 2 / 4 points; anything else: by budget
                    if (threads > 256) {
                        wps = 4;
                        if (early == 4) early = 2;
                    }
                    if (early != 0 && early != 2 && early != 4) early = wps == 3 ? 4 : kWinEarlyW4;
                    if (early == 4) wps = 3;
                    // (the profiling instantiation -- timeline stamps, ablation bits -- exists for the default shape only)
                    if ((wp.trace || wp.ablate) && (wps == 3 || early == 2))
                        return fail(MSDA_EINVAL, "fwd_win_trace / fwd_win_ablate: profiling build of the default launch shape only");
                    if constexpr (sizeof(TV) == 4) {
                        if (wide) {     // "fwd_win_wps" 2: two wavefronts per SIMD, 256 registers, twelve LDS points per wait
                            // (six points per wait, "p6", was built and measured too: 67.5 / 70.1 us against p12's 65.9 / 68.6
                            //  -- profiles/r06_fwd_win_sweep_wide.txt; not kept in the library)
                            const int e4 = opt_fwd_win_early.load() == 4;
                            if (fused) {
                                if (e4) MSDA_LAUNCH_WIN_P(true, 2, 4, 12, "msda_fwd_d32_win<fused,w2,e4,p12>");
                                else MSDA_LAUNCH_WIN_P(true, 2, 0, 12, "msda_fwd_d32_win<fused,w2,p12>");
                            } else {
                                if (e4) MSDA_LAUNCH_WIN_P(false, 2, 4, 12, "msda_fwd_d32_win<w2,e4,p12>");
                                else MSDA_LAUNCH_WIN_P(false, 2, 0, 12, "msda_fwd_d32_win<w2,p12>");
                            }
                            return check_launch(g_kernel);
                        }
                    }
                    if constexpr (sizeof(TV) == 2) {       // bf16 rows: the default shape only (512 threads, no early loads)
                        if (fused) MSDA_LAUNCH_WIN(true, 4, 0, "msda_fwd_d32_win<bf16,fused,w4>");
                        else MSDA_LAUNCH_WIN(false, 4, 0, "msda_fwd_d32_win<bf16,w4>");
                    } else {
                    if (wps == 3) {
                            if (fused) MSDA_LAUNCH_WIN(true, 3, 4, "msda_fwd_d32_win<fused,w3,e4>");
                            else MSDA_LAUNCH_WIN(false, 3, 4, "msda_fwd_d32_win<w3,e4>");
                        } else if (early == 2) {
                            if (fused) MSDA_LAUNCH_WIN(true, 4, 2, "msda_fwd_d32_win<fused,w4,e2>");
                            else MSDA_LAUNCH_WIN(false, 4, 2, "msda_fwd_d32_win<w4,e2>");
                        } else {
                            if (fused) {
                                if (wp.trace || wp.ablate) MSDA_LAUNCH_WIN_T(true, 4, 0, "msda_fwd_d32_win<fused,w4>");
                                else MSDA_LAUNCH_WIN(true, 4, 0, "msda_fwd_d32_win<fused,w4>");
                            } else {
                                if (wp.trace || wp.ablate) MSDA_LAUNCH_WIN_T(false, 4, 0, "msda_fwd_d32_win<w4>");
                                else MSDA_LAUNCH_WIN(false, 4, 0, "msda_fwd_d32_win<w4>");
                            }
                        }
                    }
#undef MSDA_LAUNCH_WIN
#undef MSDA_LAUNCH_WIN_P
#undef MSDA_LAUNCH_WIN_T
                    return check_launch(g_kernel);
                }
                variant = 3;  // the windowed kernel does not apply to this call
            }
        }
        if (variant > 4) variant = 3;
        constexpr int ROWS = RowGeom<TV>::kRows;
        int block = opt_fwd_block.load();
        if (block < 64 || block > 256 || (block & 63)) block = 256;  // kernels carry __launch_bounds__(256)
        const int wpb = block / 64;
        const int head_major = (opt_fwd_head_major.load() != 0 || sel_head_major) && (long)N * Lq >= 4096 ? 1 : 0;
        const long n_tasks = head_major ? (((long)N * Lq + ROWS - 1) / ROWS) * M : ((long)N * Lq * M + ROWS - 1) / ROWS;
        // small problems: one wave per block so every task gets its own CU slot
        int use_block = block;
        if (n_tasks < (long)kNumCU * wpb) use_block = 64;
        const int uwpb = use_block / 64;
        int grid = clamp_grid((n_tasks + uwpb - 1) / uwpb, opt_fwd_grid_mult.load());
        grid = (grid + 7) & ~7;  // whole blocks per XCD residue
        const size_t lds = (size_t)uwpb * ROWS * (2 * L * P + 1) * 16;
#define MSDA_LAUNCH_FWD(PTS, FU, NAME)                                                                               \
    do {                                                                                                             \
        g_kernel = NAME;                                                                                             \
        hipLaunchKernelGGL((msda_fwd_d32_gather<PTS, TV, FU>), dim3(grid), dim3(use_block), lds, stream, value,      \
                           shapes, lstart, src, N, S, M, L, Lq, P, out, (unsigned)value_bytes, head_major,          \
                           (unsigned)((strided ? vstride : (long)M * D) * (long)sizeof(TV)));                         \
    } while (0)
        // (four points = 16 corner rows in flight per lane: the best of the round-1 sweep, profiles/r01_kbench_fwd_sweep.txt;
        //  the 1- and 2-point instantiations went in round 5 -- variants 2, 3, 4 all mean this kernel)
        if (fused) MSDA_LAUNCH_FWD(4, true, sizeof(TV) == 2 ? "msda_fwd_d32_gather<4,bf16,fused>" : "msda_fwd_d32_gather<4,fused>");
        else MSDA_LAUNCH_FWD(4, false, sizeof(TV) == 2 ? "msda_fwd_d32_gather<4,bf16>" : "msda_fwd_d32_gather<4>");
#undef MSDA_LAUNCH_FWD
        return check_launch(g_kernel);
    }
    return fail(MSDA_ENOTSUP, "no specialised forward for this dtype");
}

template <typename TV, typename TC, typename TG>
int backward_impl(const TV *value, const int64_t *shapes, const int64_t *lstart, const TC *loc, const TC *attn,
                  const FusedArgs &fa, const TV *grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                  TG *grad_value, TC *grad_loc, TC *grad_attn, float *grad_proj, float *grad_ref_part,
                  int zero_grad_value, const int64_t *shapes_host, hipStream_t stream, float *workspace = nullptr,
                  size_t workspace_bytes = 0, const TV *fwd_out = nullptr) {
    const long vstride = take_value_stride();
    const bool fused = fa.proj != nullptr;
    // `fwd_out` (round 6): the forward's output of the same call, when the caller still holds it.  sum_j a_j ga_j of the
    // softmax Jacobian IS <grad_out_row, out_row>, so with it the counting-sort backward needs no side kernel
    const bool soft_ok = fused && fwd_out != nullptr && sizeof(TV) == 4 && L * P == 16 && fa.ref_dim == 2 &&
                         (fa.proj_stride % 4) == 0 && ((2 * M * L * P) % 4) == 0 && (((uintptr_t)fa.proj) & 15) == 0 &&
                         L == 4 && P == 4 && grad_ref_part == nullptr && opt_bwd_soft.load() != 0;
    int rc = fused ? check_dims(value, shapes, lstart, fa.proj, fa.ref, grad_out, N, S, M, D, L, Lq, P)
                   : check_dims(value, shapes, lstart, loc, attn, grad_out, N, S, M, D, L, Lq, P);
    if (rc) return rc;
    if (fused) {
        if ((rc = check_fused(fa, M, L, P))) return rc;
        if (!grad_value || !grad_proj) return fail(MSDA_EINVAL, "null gradient pointer");
    } else if (!grad_value || !grad_loc || !grad_attn) {
        return fail(MSDA_EINVAL, "null gradient pointer");
    }
    // grad_value is accumulated into: zeroed here on request -- by a kernel, not a memset node (a replayed hipGraph of
    // ROCm 7.2 does not order MEMSET nodes behind the kernels before them unless DEBUG_CLR_GRAPH_PACKET_CAPTURE=0,
    // tools/graph_memset_probe.py) -- together with whatever else the chosen path wants cleared (the sorted backward's
    // bucket totals: one launch instead of two)
    const size_t gv_words = (size_t)N * S * M * D * (sizeof(TG) / 4);
    auto zero_launch = [&](unsigned *extra, unsigned extra_n) -> int {
        if (!zero_grad_value && extra_n == 0u) return MSDA_OK;
        const size_t n1 = zero_grad_value ? gv_words : 0;
        const size_t blocks = (n1 / 4 + extra_n + 255) / 256;
        hipLaunchKernelGGL(msda_zero_words_kernel, dim3((unsigned)(blocks < 2048 ? (blocks ? blocks : 1) : 2048)), dim3(256), 0, stream,
                           reinterpret_cast<unsigned *>(grad_value), n1, extra, extra_n);
        return check_launch("msda_zero_words_kernel");
    };
    // `value` and `grad_value` as slices of wider tensors (msda_next_value_pixel_stride): the rows kernel only, and the
    // caller owns the zeroing of the whole gradient tensor
    const bool strided = vstride != 0 && vstride != (long)M * D;
    if (strided && zero_grad_value) return fail(MSDA_EINVAL, "value pixel stride: grad_value is the caller's to zero");
    if ((long)N * Lq == 0) return zero_launch(nullptr, 0u);
    int variant = opt_bwd_variant.load();
    const long value_elems = strided ? (long)N * S * vstride : (long)N * S * M * D;
    const long value_bytes = value_elems * (long)sizeof(TV);
    constexpr bool kD32Type = sizeof(TC) == 4 && sizeof(TG) == 4 && (sizeof(TV) == 4 || sizeof(TV) == 2);
    if (strided) {
        if (vstride < (long)M * D || (vstride * (long)sizeof(TV)) % 16 != 0)
            return fail(MSDA_EINVAL, "value pixel stride: at least M * D elements, rows 16-byte aligned");
        if (!(kD32Type && d32_ok(D, L, value_elems) && L * P <= kRowsMaxLP && (long)N * Lq * M < (1L << 31)))
            return fail(MSDA_ENOTSUP, "value pixel stride: D = 32 float32 / bfloat16 calls of at most 64 points only");
        variant = 3;        // (any value the pyramid / sorted branches do not claim: straight to the rows kernel)
    }
    const bool can_tile = !strided && kD32Type && d32_ok(D, L, value_elems) && shapes_host && Lq == S && L <= kTileMaxL &&
                          grad_ref_part == nullptr;
    // Measured on MI355X (profiles/): per-contribution global float atomics cap the backward at ~1.1 ms
    // for the encoder call (L2 atomic throughput; the row-per-block kernel's 32-consecutive-lane pattern
    // is the fastest of them).  Self-attention over the pyramid (Lq == S, host shapes known) therefore
    // takes the region-tiled kernel that pre-reduces grad_value in fixed-point LDS windows; every other call
    // (decoder queries) takes the row-per-block kernel.
    // Self-attention over the pyramid: the counting-sort kernel (msda_bwd_bins.h, round 4: 145 us at the encoder
    // shape; tile_lv 218, tile_q2 317), at the window margin the measured off-window share asks for -- or, when most
    // points leave even the large window (uniformly random locations), no windows at all (msda_select.h).
    SelSlot *slot = nullptr;
    int sel = 0, bins_margin = opt_bwd_bins_margin.load(), bins_shrink = 0;
    if (kD32Type && can_tile && (variant == 0 || variant == 12)) {
        slot = sel_acquire(1, M, L, P, (int)sizeof(TV), stream);
        bool probe = false;
        sel = sel_level(slot, 1, probe, slot != nullptr && stream_capturing(stream));
        if (variant == 12) sel = opt_sel_level.load() >= 0 ? sel : 0;         // forced: the selector only measures
        if (sel >= 1) {
            bins_shrink = opt_bwd_bins_margin_hi.load() - bins_margin;
            bins_margin = opt_bwd_bins_margin_hi.load();
            if (bins_shrink < 0) bins_shrink = 0;
        }
        if (variant == 0) variant = sel >= 2 ? (opt_bwd_sorted.load() ? 13 : 1) : 12;
    }
    // ---- variant 13: grad_value by sort + gather through the caller's scratch (msda_bwd_sorted.h); everything else of
    //      the call from msda_bwd_d32_rows without its atomics.  Not tied to the pyramid: any D = 32 call the rows kernel
    //      takes.  Too little scratch (or none): the rows kernel with its atomics, below ----
    if constexpr (kD32Type) {
        if (variant == 13) {
            const long n_rows13 = (long)N * Lq * M;
            const size_t fused_bytes = fused ? (((size_t)n_rows13 * L * P * 3 * sizeof(float) + 255) & ~(size_t)255) : 0;
            SortPlan sp;
            const bool fits = d32_ok(D, L, value_elems) && L * P <= kRowsMaxLP && n_rows13 < (1L << 31) &&
                              workspace != nullptr && workspace_bytes > fused_bytes &&
                              make_sort_plan(sp, N, S, M, L, Lq, P, sizeof(TV), (unsigned char *)workspace + fused_bytes,
                                             opt_bwd_sort_qc.load(), opt_bwd_sort_emult.load()) &&
                              fused_bytes + sp.bytes <= workspace_bytes &&
                              sort_dots_lds(sp.qc, L * P, sp.nbk).bytes <= 64u * 1024u;
            if (fits) {
                // grad_value (on request) and the bucket totals start from zero: one launch
                if ((rc = zero_launch(sp.cursor, (unsigned)((size_t)N * M * sp.nbk)))) return rc;
                const PointSrc src = make_src(loc, attn, fa, M, L, P);
                const int threads = 256, per = threads / 32;
                const int rgrid = clamp_grid((n_rows13 + per - 1) / per, 64);
                const unsigned gv_bytes = (unsigned)(value_elems * 4);
                const unsigned n_cursor = (unsigned)((size_t)N * M * sp.nbk);
                const bool b16 = sizeof(TV) == 2;
                // 1. fused: the softmax weights once per row into the scratch; the locations too unless the kernels below
                //    can compute them from the raw projection themselves (the slim form of the split backward: L * P = 16,
                //    16-byte aligned rows -- one lane per row in both side kernels)
                PointSrc src_k = src;
                int fused_loc = 0, offsets_done = 0, soft16 = 0;
                bool slim = false;
                if (fused) {
                    float *loc_ws = workspace, *attn_ws = workspace + (size_t)n_rows13 * L * P * 2;
                    slim = L * P == 16 && (fa.proj_stride % 4) == 0 && (src.n_off % 4) == 0 && grad_ref_part == nullptr &&
                           (((uintptr_t)fa.proj | (uintptr_t)grad_proj | (uintptr_t)workspace) & 15) == 0 &&
                           opt_bwd_side_rows.load() != 0;
                    if (slim) {
                        // L * P = 16, 2-d reference points (the encoder's): the sixteen lanes of a row compute its softmax
                        // in the dots and emit kernels themselves (one function, the bits of msda_fused_attn16_rows_kernel)
                        // and the dots kernel applies the softmax Jacobian -- no side kernel, no weights in HBM.  4-d
                        // reference points keep the weights in the scratch for the finishing kernel's location Jacobian
                        fused_loc = 1;
                        offsets_done = fa.ref_dim == 2 ? 1 : 0;
                        soft16 = offsets_done;
                        if (!soft16)
                            hipLaunchKernelGGL(msda_fused_attn16_rows_kernel, dim3(clamp_grid((n_rows13 + 255) / 256, 32)),
                                               dim3(256), 0, stream, src, (unsigned)n_rows13, (unsigned)M, attn_ws);
                    } else if (L * P <= 16) {
                        hipLaunchKernelGGL(msda_fused_points16_kernel, dim3(clamp_grid((n_rows13 * 16 + 255) / 256, 32)),
                                           dim3(256), 0, stream, shapes, src, n_rows13, M, L, P, loc_ws, attn_ws);
                    } else {
                        hipLaunchKernelGGL(msda_fused_points_kernel, dim3(clamp_grid((n_rows13 * 8 + 255) / 256, 16)),
                                           dim3(256), 0, stream, shapes, src, n_rows13, M, L, P, loc_ws, attn_ws);
                    }
                    if ((rc = check_launch("msda_fused_points_kernel"))) return rc;
                    src_k.loc = loc_ws;
                    src_k.attn = attn_ws;
                }
                const unsigned go_bytes = (unsigned)((size_t)n_rows13 * 32 * sizeof(TV));
                // 2. the bucket totals start from zero; then, per (batch, head, chunk of queries): grad_loc / grad_attn
                //    (fused: the columns of grad_proj) and the chunk's corner histogram
                const SortDotsLds dl = sort_dots_lds(sp.qc, L * P, sp.nbk);
                const int pgrid = (N * M * sp.nchunk + 7) & ~7;
                // (with gradients of the reference points wanted -- no caller of this package asks for them on this path --
                //  the rows kernel without its atomics runs AFTER and overwrites the columns of grad_proj with its own,
                //  finished, results next to grad_ref_part; the dots kernel then only counts)
                hipLaunchKernelGGL((msda_bwd_sort_dots<TV>), dim3(pgrid), dim3(kSortThreads), dl.bytes, stream, value, shapes,
                                   lstart, src_k, fused_loc, offsets_done, soft16, grad_out, (float *)grad_loc, (float *)grad_attn,
                                   fused ? grad_proj : (float *)nullptr, sp, dl, (unsigned)value_bytes);
                if ((rc = check_launch("msda_bwd_sort_dots"))) return rc;
                if (fused && grad_ref_part != nullptr) {
                    hipLaunchKernelGGL((msda_bwd_d32_rows<TV, true, false>), dim3(rgrid), dim3(threads), 0, stream, value, shapes,
                                       lstart, src, grad_out, N, S, M, L, Lq, P, (float *)grad_value, (float *)nullptr,
                                       (float *)nullptr, grad_proj, grad_ref_part, (unsigned)value_bytes, gv_bytes,
                                       (unsigned *)nullptr, 0u);
                    if ((rc = check_launch("msda_bwd_d32_rows<no atomics>"))) return rc;
                } else if (fused && !soft16) {       // (soft16: finished by the dots kernel)
                    if (L * P <= 16)
                        hipLaunchKernelGGL(msda_fused_finish16_kernel, dim3(clamp_grid((n_rows13 * 16 + 255) / 256, 32)),
                                           dim3(256), 0, stream, shapes, src_k, n_rows13, M, L, P, grad_proj, offsets_done);
                    else
                        hipLaunchKernelGGL(msda_fused_finish_kernel, dim3(clamp_grid((n_rows13 * 8 + 255) / 256, 16)),
                                           dim3(256), 0, stream, shapes, src_k, n_rows13, M, L, P, grad_proj);
                    if ((rc = check_launch("msda_fused_finish_kernel"))) return rc;
                }
                // 3. scan -> emit -> gather
                const size_t hist_lds = (size_t)sp.nbk * 4;
                const int egrid = (N * M * ((sp.nchunk + sp.emult - 1) / sp.emult) + 7) & ~7;
                hipLaunchKernelGGL(msda_bwd_sort_scan, dim3(N * M), dim3(kSortThreads), 0, stream, sp);
                hipLaunchKernelGGL(msda_bwd_sort_emit, dim3(egrid), dim3(kSortThreads), hist_lds, stream, shapes, lstart, src_k,
                                   fused_loc, soft16, sp);
                if ((rc = check_launch("msda_bwd_sort_emit"))) return rc;
                const int ggrid = (N * M * sp.max_items + 7) & ~7;
                const size_t glds = (size_t)(kSortSlice + 2) * 8 + (size_t)kSortBP * 8;
                g_kernel = fused ? (b16 ? "msda_bwd_d32_sorted<bf16,fused>" : "msda_bwd_d32_sorted<fused>")
                                 : (b16 ? "msda_bwd_d32_sorted<bf16>" : "msda_bwd_d32_sorted");
                hipLaunchKernelGGL((msda_bwd_sort_gather<TV>), dim3(ggrid), dim3(kSortThreads), glds, stream, grad_out,
                                   (float *)grad_value, sp, go_bytes);
                if ((rc = check_launch(g_kernel))) return rc;
                hipLaunchKernelGGL(msda_bwd_sort_reduce, dim3((N * M * sp.nbk + 7) & ~7), dim3(kSortThreads), 0, stream,
                                   (float *)grad_value, sp);
                {
                    const char *name = g_kernel;
                    rc = check_launch("msda_bwd_sort_reduce");
                    g_kernel = name;
                }
                return rc;
            }
            variant = 1;        // (-> msda_bwd_d32_rows with its atomics, below)
        }
    }
    if (variant == 13) variant = 1;
    if ((rc = zero_launch(nullptr, 0u))) return rc;
    if (variant == 0 || (variant >= 2 && variant != 10 && variant != 12)) variant = can_tile && P <= 8 ? 10 : 1;
    if (variant >= 2 && !can_tile) variant = 1;
    if constexpr (kD32Type) {
        if (variant == 12) {        // counting-sort gather (msda_bwd_bins.h); the one-kernel fused form stays with tile_lv
            const bool will_split = fa.proj != nullptr && opt_bwd_split.load() != 0 &&
                                    (soft_ok || (workspace != nullptr &&
                                                 workspace_bytes >= (size_t)N * Lq * M * L * P * 3 * sizeof(float)));
            if (P > 8 || (fa.proj != nullptr && !will_split)) variant = 10;
        }
        if (variant == 10 || variant == 12) {       // one pyramid level per workgroup
            TilePlan pl;
            size_t lds_all = 0;
            BinsPlan bp;
            memset(&bp, 0, sizeof(bp));
            int bins_ni = 0;
            size_t bins_lds = 0;
            bool planned = false;
            if (variant == 12) {
                if (make_tile_plan(pl, shapes_host, N, S, M, D, L, Lq, P, value_bytes, bins_margin, 0, 8, 0, lds_all, false)) {
                    for (int ni = 2; ni <= 3 && !bins_ni; ++ni)
                        if (make_bins_plan(bp, pl, ni, bins_lds)) bins_ni = ni;
                }
                planned = bins_ni != 0;
                if (!planned) variant = 10;
                bp.shrink = bins_shrink;
                bp.level = sel;
                // (cumulative counters, fixed addresses: a captured launch counts like an eager one)
                bp.stats = slot ? slot->dev : nullptr;
                bp.stats_host = slot ? slot->host_dev : nullptr;
            }
            if (!planned)
                planned = P <= 8 && make_tile_plan(pl, shapes_host, N, S, M, D, L, Lq, P, value_bytes,
                                                   opt_bwd_tile_margin.load(), 0, 8, 0, lds_all);
            if (planned) {
                int win_max = 0;
                for (int l = 0; l < L; ++l) win_max = pl.win[l] > win_max ? pl.win[l] : win_max;
                const size_t lds = variant == 12 ? bins_lds
                                                 : (size_t)(win_max * win_max + 8) * 128 + (size_t)32 * (2 * P + 1) * 16;
                const int grid = (pl.n_blocks * L + 7) & ~7;
                PointSrc src = make_src(loc, attn, fa, M, L, P);
                pl.ablate = opt_bwd_ablate.load();
                pl.wide_log2 = opt_bwd_wide_log2.load();
                const long n_rows = (long)N * Lq * M;
                // Split fused backward (needs the caller's workspace): materialise the prologue once -- the tiled
                // kernel would otherwise redo the row softmax and the location arithmetic in each of its L
                // workgroups per region -- run the plain kernel on it, finish the Jacobians in place.
                const bool soft = soft_ok && variant == 12 && opt_bwd_split.load() != 0;      // (no workspace needed)
                const bool split = fused && opt_bwd_split.load() != 0 &&
                                   (soft || (workspace != nullptr && workspace_bytes >= (size_t)n_rows * L * P * 3 * sizeof(float)));
                // The counting-sort kernel computes the locations itself (one lane per point: the arithmetic is
                // cheap there) and, for 2-d reference points, writes the final offset gradients: the two side
                // kernels then move a third of the bytes (attention weights out, the softmax Jacobian in place).
                const bool slim = split && variant == 12 && L * P <= 16;
                const int offsets_done = slim && fa.ref_dim == 2 ? 1 : 0;
                bp.fused_loc = slim ? 1 : 0;
                bp.offsets_done = offsets_done;
                bool rows16 = false;
                bp.soft = soft ? 1 : 0;
                if (split && !soft) {
                    float *loc_ws = workspace, *attn_ws = workspace + (size_t)n_rows * L * P * 2;
                    // one lane per row (16 points as four 16-byte accesses): the slim path's two side kernels
                    rows16 = slim && L * P == 16 && (fa.proj_stride % 4) == 0 && (src.n_off % 4) == 0 &&
                             (((uintptr_t)fa.proj | (uintptr_t)grad_proj | (uintptr_t)workspace) & 15) == 0 &&
                             n_rows < (1L << 31) && opt_bwd_side_rows.load() != 0;
                    if (rows16) {
                        hipLaunchKernelGGL(msda_fused_attn16_rows_kernel, dim3(clamp_grid((n_rows + 255) / 256, 32)),
                                           dim3(256), 0, stream, src, (unsigned)n_rows, (unsigned)M, attn_ws);
                    } else if (L * P <= 16) {
                        hipLaunchKernelGGL(msda_fused_points16_kernel, dim3(clamp_grid((n_rows * 16 + 255) / 256, 32)),
                                           dim3(256), 0, stream, shapes, src, n_rows, M, L, P,
                                           slim ? (float *)nullptr : loc_ws, attn_ws);
                    } else {
                        hipLaunchKernelGGL(msda_fused_points_kernel, dim3(clamp_grid((n_rows * 8 + 255) / 256, 16)),
                                           dim3(256), 0, stream, shapes, src, n_rows, M, L, P, loc_ws, attn_ws);
                    }
                    if ((rc = check_launch("msda_fused_points_kernel"))) return rc;
                    src.loc = loc_ws;
                    src.attn = attn_ws;
                }
#define MSDA_LAUNCH_LV(PTS, FU, NAME)                                                                                \
    do {                                                                                                             \
        rc = allow_big_lds(msda_bwd_d32_tile_lv<PTS, TV, FU>, lds);                                                  \
        if (rc) return rc;                                                                                           \
        g_kernel = NAME;                                                                                             \
        hipLaunchKernelGGL((msda_bwd_d32_tile_lv<PTS, TV, FU>), dim3(grid), dim3(kTileThreads), lds, stream, value,  \
                           lstart, src, grad_out, (float *)grad_value, (float *)grad_loc, (float *)grad_attn,        \
                           grad_proj, pl);                                                                           \
    } while (0)
                const bool b16 = sizeof(TV) == 2;
#define MSDA_LAUNCH_BINS(NI, NAME)                                                                                   \
    do {                                                                                                             \
        g_kernel = NAME;                                                                                             \
        hipLaunchKernelGGL((msda_bwd_d32_bins<NI, TV>), dim3(grid), dim3(kTileThreads), lds, stream, value, lstart,  \
                           src, grad_out, (float *)grad_value, (float *)grad_loc, (float *)grad_attn, grad_proj, pl, \
                           bp, (const TV *)nullptr);                                                                 \
    } while (0)
                if (variant == 12 && soft) {       // (fp32, L = P = 4) everything of the fused backward in the one kernel
                    if constexpr (sizeof(TV) == 4) {
                        g_kernel = bins_ni == 2 ? "msda_bwd_d32_tile_bins<split,soft>" : "msda_bwd_d32_tile_bins<3,split,soft>";
                        if (bins_ni == 2)
                            hipLaunchKernelGGL((msda_bwd_d32_bins<2, TV, true>), dim3(grid), dim3(kTileThreads), lds, stream, value,
                                               lstart, src, grad_out, (float *)grad_value, (float *)grad_loc, (float *)grad_attn,
                                               grad_proj, pl, bp, fwd_out);
                        else
                            hipLaunchKernelGGL((msda_bwd_d32_bins<3, TV, true>), dim3(grid), dim3(kTileThreads), lds, stream, value,
                                               lstart, src, grad_out, (float *)grad_value, (float *)grad_loc, (float *)grad_attn,
                                               grad_proj, pl, bp, fwd_out);
                    }
                } else if (variant == 12) {
                    if (bins_ni == 2) MSDA_LAUNCH_BINS(2, split ? (b16 ? "msda_bwd_d32_tile_bins<bf16,split>" : "msda_bwd_d32_tile_bins<split>")
                                                                : (b16 ? "msda_bwd_d32_tile_bins<bf16>" : "msda_bwd_d32_tile_bins"));
                    else MSDA_LAUNCH_BINS(3, split ? (b16 ? "msda_bwd_d32_tile_bins<3,bf16,split>" : "msda_bwd_d32_tile_bins<3,split>")
                                                   : (b16 ? "msda_bwd_d32_tile_bins<3,bf16>" : "msda_bwd_d32_tile_bins<3>"));
                } else {
                    if (split) MSDA_LAUNCH_LV(2, false, b16 ? "msda_bwd_d32_tile_lv<2,bf16,split>" : "msda_bwd_d32_tile_lv<2,split>");
                    else if (fused) MSDA_LAUNCH_LV(2, true, b16 ? "msda_bwd_d32_tile_lv<2,bf16,fused>" : "msda_bwd_d32_tile_lv<2,fused>");
                    else MSDA_LAUNCH_LV(2, false, b16 ? "msda_bwd_d32_tile_lv<2,bf16>" : "msda_bwd_d32_tile_lv<2>");
                }
#undef MSDA_LAUNCH_LV
#undef MSDA_LAUNCH_BINS
                rc = check_launch(g_kernel);
                if (rc || !fused || soft) return rc;
                const int jgrid = clamp_grid((n_rows * 8 + 255) / 256, 16);
                if (split) {
                    if (rows16 && offsets_done) {
                        hipLaunchKernelGGL(msda_fused_finish16_rows_kernel, dim3(clamp_grid((n_rows + 255) / 256, 32)),
                                           dim3(256), 0, stream, src, (unsigned)n_rows, (unsigned)M, grad_proj);
                    } else if (L * P <= 16) {
                        hipLaunchKernelGGL(msda_fused_finish16_kernel, dim3(clamp_grid((n_rows * 16 + 255) / 256, 32)),
                                           dim3(256), 0, stream, shapes, src, n_rows, M, L, P, grad_proj, offsets_done);
                    } else {
                        hipLaunchKernelGGL(msda_fused_finish_kernel, dim3(jgrid), dim3(256), 0, stream, shapes, src,
                                           n_rows, M, L, P, grad_proj);
                    }
                    const char *name = g_kernel;
                    rc = check_launch("msda_fused_finish_kernel");
                    g_kernel = name;
                    return rc;
                }
                hipLaunchKernelGGL(msda_softmax_jacobian_kernel, dim3(jgrid), dim3(256), 0, stream, src, n_rows, M, L * P,
                                   grad_proj);
                const char *name = g_kernel;
                rc = check_launch("msda_softmax_jacobian_kernel");
                g_kernel = name;
                return rc;
            }
            variant = 1;
        }
    }
    const long n_rows = (long)N * Lq * M;
    if constexpr (kD32Type) {
        // few-query calls at D = 32 (the decoder's cross-attention): 32 lanes per row, grad_value atomics in whole
        // 128-byte rows, four points in flight (msda_bwd_rows.h)
        const int requested = opt_bwd_variant.load();      // 1 = the generic kernel, explicitly
        if (d32_ok(D, L, value_elems) && L * P <= kRowsMaxLP && (opt_bwd_rows.load() != 0 || strided) && n_rows < (1L << 31) &&
            (requested != 1 || strided)) {
            const PointSrc src = make_src(loc, attn, fa, M, L, P);
            // a row is a chain of dependent round trips (stage -> loads -> atomics -> reductions): small calls get one
            // wavefront (two rows) per workgroup so that every CU holds several chains
            int threads = opt_bwd_rows_block.load();
            if (threads != 64 && threads != 128 && threads != 256) threads = n_rows <= 16384 ? 64 : 256;
            const int per = threads / 32;
            const int grid = clamp_grid((n_rows + per - 1) / per, 64);
            const unsigned gv_bytes = (unsigned)(value_elems * 4);
            const bool b16 = sizeof(TV) == 2;
            if (fused) {
                g_kernel = b16 ? "msda_bwd_d32_rows<bf16,fused>" : "msda_bwd_d32_rows<fused>";
                hipLaunchKernelGGL((msda_bwd_d32_rows<TV, true>), dim3(grid), dim3(threads), 0, stream, value, shapes, lstart,
                                   src, grad_out, N, S, M, L, Lq, P, (float *)grad_value, (float *)nullptr,
                                   (float *)nullptr, grad_proj, grad_ref_part, (unsigned)value_bytes, gv_bytes,
                                   (unsigned *)nullptr, 0u, (unsigned)(strided ? vstride : 0));
            } else {
                g_kernel = b16 ? "msda_bwd_d32_rows<bf16>" : "msda_bwd_d32_rows";
                hipLaunchKernelGGL((msda_bwd_d32_rows<TV, false>), dim3(grid), dim3(threads), 0, stream, value, shapes, lstart,
                                   src, grad_out, N, S, M, L, Lq, P, (float *)grad_value, (float *)grad_loc,
                                   (float *)grad_attn, (float *)nullptr, (float *)nullptr, (unsigned)value_bytes, gv_bytes,
                                   (unsigned *)nullptr, 0u, (unsigned)(strided ? vstride : 0));
            }
            return check_launch(g_kernel);
        }
    }
    int block = ((D + 63) / 64) * 64;
    if (block > 1024) block = 1024;
    const int grid = (int)(n_rows < 65536L * 16 ? n_rows : 65536L * 16);
    if constexpr (sizeof(TC) == 4) {
        if (fused) {
            const PointSrc src = make_src(loc, attn, fa, M, L, P);
            g_kernel = "msda_bwd_generic<fused>";
            hipLaunchKernelGGL((msda_bwd_generic<TV, TC, TG, true>), dim3(grid), dim3(block), 0, stream, value, shapes,
                               lstart, (const TC *)nullptr, (const TC *)nullptr, src, grad_out, N, S, M, D, L, Lq, P,
                               grad_value, (TC *)nullptr, (TC *)nullptr, grad_proj, grad_ref_part);
            return check_launch(g_kernel);
        }
    }
    g_kernel = "msda_bwd_generic";
    hipLaunchKernelGGL((msda_bwd_generic<TV, TC, TG, false>), dim3(grid), dim3(block), 0, stream, value, shapes, lstart,
                       loc, attn, PointSrc{}, grad_out, N, S, M, D, L, Lq, P, grad_value, grad_loc, grad_attn,
                       (float *)nullptr, (float *)nullptr);
    return check_launch(g_kernel);
}

}  // namespace

extern "C" {

int msda_abi_version(void) { return 7; }

void msda_set_call_site(uint64_t site) { g_site = site; }

int msda_next_value_pixel_stride(long elements) {
    if (elements < 0) return fail(MSDA_EINVAL, "msda_next_value_pixel_stride: negative stride");
    g_value_stride = elements;
    return MSDA_OK;
}

int msda_selector_last(int *level, float *off_share, float *inner_share) {
    if (level) *level = g_sel_level;
    if (off_share) *off_share = g_sel_frac;
    if (inner_share) *inner_share = g_sel_inner;
    return MSDA_OK;
}

int msda_selector_next(int kind, int level, int off_permille, int inner_permille) {
    return sel_next_level(kind, level, (float)off_permille, (float)inner_permille, sel_rule(kind));
}

static int selector_poll_impl(const uint64_t *sites, int n_sites, int probe, uint64_t *signature) {
    int dev = 0, n = 0;
    (void)hipGetDevice(&dev);
    unsigned long long h = 0ull;
    const int pinned = opt_sel_level.load();
    if (opt_auto_select.load()) {
        std::lock_guard<std::mutex> lock(g_sel_mu);
        // (unfiltered: every kSelProbeEvery-th poll of the process is a probe tick; filtered: the caller counts its own)
        const bool probe_tick = sites ? probe != 0 : (++g_sel_tick % kSelProbeEvery) == 0ull;
        for (int i = 0; i < kSelSlots; ++i) {
            SelSlot &s = g_sel[i];
            if (!s.used || s.key.dev != dev) continue;
            if (sites) {
                bool mine = false;
                for (int k = 0; k < n_sites && !mine; ++k) mine = sites[k] == s.key.site;
                if (!mine) continue;
            }
            // launches are arriving (replayed graphs make no library call): the record is in use, whatever its stamp says
            if (s.host[9] != s.pub_seen) { s.pub_seen = s.host[9]; s.stamp = ++g_sel_clock; }
            sel_refresh(&s);
            const int top = s.key.kind == 0 ? 1 : 2;
            s.eff = (s.level == top && probe_tick) ? top - 1 : s.level;
            s.polled = true;
            const int level = pinned >= 0 ? pinned : s.eff;
            if (level != 0) {       // (records at level 0 -- the state before anything was measured -- leave no mark)
                // (site, direction, element size, level) -- not the slot index: the same levels give the same signature
                //  after the least-recently-used rule has moved a site to another record.  Order-independent (a sum of
                //  per-record hashes), for the same reason
                const unsigned long long w[2] = {s.key.site ^ ((unsigned long long)s.key.kind << 63) ^ ((unsigned long long)s.key.dt << 56),
                                                 (unsigned long long)level};
                unsigned long long hr = 0xcbf29ce484222325ull;
                for (int k = 0; k < 2; ++k)
                    for (int b = 0; b < 8; ++b) hr = (hr ^ ((w[k] >> (8 * b)) & 0xffull)) * 0x100000001b3ull;
                h += hr | 1ull;
            }
            ++n;
        }
    }
    if (signature) *signature = h;
    return n;
}

// For callers that REPLAY captured launches (no library call per launch): read every record of the current device,
// move the levels, and return a signature of the levels a call would run at now (0 when nothing is selected).  Every
// kSelProbeEvery-th poll announces one level down for the records that sit at a level without windows -- the graph
// captured under that signature is the probe.  Returns the number of records read.
int msda_selector_poll(uint64_t *signature) { return selector_poll_impl(nullptr, 0, 0, signature); }

// The same for the records of the given call sites only (ABI 6): a graph cache hashes the levels of the modules its
// graphs hold -- a level move or a probe of some other module's record no longer changes its key (advisor, round 5) --
// and decides itself when its records probe (`probe` != 0: records at a level without windows announce one level down).
int msda_selector_poll_sites(const uint64_t *sites, int n_sites, int probe, uint64_t *signature) {
    if (n_sites < 0 || (n_sites > 0 && !sites)) return fail(MSDA_EINVAL, "msda_selector_poll_sites: null site list");
    static const uint64_t none = 0;
    return selector_poll_impl(sites ? sites : &none, n_sites, probe, signature);
}
int msda_selector_reset(void) {
    std::lock_guard<std::mutex> lock(g_sel_mu);
    const size_t dbytes = (size_t)kSelSlots * kSelDevWords * 8, hbytes = (size_t)kSelSlots * kSelHostWords * 8;
    int cur = 0;
    (void)hipGetDevice(&cur);
    int rc = MSDA_OK;
    for (int d = 0; d < 64; ++d) {
        if (!g_sel_pool_dev[d]) continue;
        // (launches in flight still count into the block: wait for them, then clear it)
        if (hipSetDevice(d) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
            hipMemset(g_sel_pool_dev[d], 0, dbytes) != hipSuccess) {
            const hipError_t e = hipGetLastError();
            rc = fail((int)(e != hipSuccess ? e : hipErrorUnknown), "msda_selector_reset: device block not cleared");
            continue;
        }
        memset(g_sel_pool_host[d], 0, hbytes);
    }
    (void)hipSetDevice(cur);
    for (int i = 0; i < kSelSlots; ++i) g_sel[i].used = false;
    g_sel_level = 0;
    g_sel_frac = g_sel_inner = -1.f;
    return rc;
}
const char *msda_last_error(void) { return g_err; }
const char *msda_last_kernel(void) { return g_kernel; }

int msda_forward_f32(const float *value, const int64_t *shapes_dev, const int64_t *lstart_dev, const float *loc,
                     const float *attn, int N, int S, int M, int D, int L, int Lq, int P, float *out,
                     const int64_t *shapes_host, void *stream) {
    return forward_impl<float, float>(value, shapes_dev, lstart_dev, loc, attn, FusedArgs{}, N, S, M, D, L, Lq, P, out,
                                      shapes_host, (hipStream_t)stream);
}

int msda_forward_f64(const double *value, const int64_t *shapes_dev, const int64_t *lstart_dev, const double *loc,
                     const double *attn, int N, int S, int M, int D, int L, int Lq, int P, double *out,
                     const int64_t *shapes_host, void *stream) {
    return forward_impl<double, double>(value, shapes_dev, lstart_dev, loc, attn, FusedArgs{}, N, S, M, D, L, Lq, P,
                                        out, shapes_host, (hipStream_t)stream);
}

int msda_forward_bf16(const uint16_t *value, const int64_t *shapes_dev, const int64_t *lstart_dev, const float *loc,
                      const float *attn, int N, int S, int M, int D, int L, int Lq, int P, uint16_t *out,
                      const int64_t *shapes_host, void *stream) {
    return forward_impl<bf16_t, float>((const bf16_t *)value, shapes_dev, lstart_dev, loc, attn, FusedArgs{}, N, S, M,
                                       D, L, Lq, P, (bf16_t *)out, shapes_host, (hipStream_t)stream);
}

int msda_backward_f32(const float *value, const int64_t *shapes_dev, const int64_t *lstart_dev, const float *loc,
                      const float *attn, const float *grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                      float *grad_value, float *grad_loc, float *grad_attn, int zero_grad_value,
                      const int64_t *shapes_host, void *stream) {
    return backward_impl<float, float, float>(value, shapes_dev, lstart_dev, loc, attn, FusedArgs{}, grad_out, N, S, M,
                                              D, L, Lq, P, grad_value, grad_loc, grad_attn, nullptr, nullptr,
                                              zero_grad_value, shapes_host, (hipStream_t)stream);
}

int msda_backward_f64(const double *value, const int64_t *shapes_dev, const int64_t *lstart_dev, const double *loc,
                      const double *attn, const double *grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                      double *grad_value, double *grad_loc, double *grad_attn, int zero_grad_value,
                      const int64_t *shapes_host, void *stream) {
    return backward_impl<double, double, double>(value, shapes_dev, lstart_dev, loc, attn, FusedArgs{}, grad_out, N, S,
                                                 M, D, L, Lq, P, grad_value, grad_loc, grad_attn, nullptr, nullptr,
                                                 zero_grad_value, shapes_host, (hipStream_t)stream);
}

int msda_backward_bf16(const uint16_t *value, const int64_t *shapes_dev, const int64_t *lstart_dev, const float *loc,
                       const float *attn, const uint16_t *grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                       float *grad_value, float *grad_loc, float *grad_attn, int zero_grad_value,
                       const int64_t *shapes_host, void *stream) {
    return backward_impl<bf16_t, float, float>((const bf16_t *)value, shapes_dev, lstart_dev, loc, attn, FusedArgs{},
                                               (const bf16_t *)grad_out, N, S, M, D, L, Lq, P, grad_value, grad_loc,
                                               grad_attn, nullptr, nullptr, zero_grad_value, shapes_host,
                                               (hipStream_t)stream);
}

// ---- fused prologue entry points ----
static FusedArgs fused_args(const float *proj, int proj_stride, const float *ref, int ref_dim, const uint8_t *mask) {
    FusedArgs fa;
    fa.proj = proj;
    fa.proj_stride = proj_stride;
    fa.ref = ref;
    fa.ref_dim = ref_dim;
    fa.mask = mask;
    return fa;
}

int msda_fused_forward_f32(const float *value, const int64_t *shapes_dev, const int64_t *lstart_dev, const float *proj,
                           int proj_stride, const float *ref, int ref_dim, const uint8_t *pad_mask, int N, int S, int M,
                           int D, int L, int Lq, int P, float *out, const int64_t *shapes_host, void *stream) {
    if (!proj) return fail(MSDA_EINVAL, "null pointer argument");
    return forward_impl<float, float>(value, shapes_dev, lstart_dev, nullptr, nullptr,
                                      fused_args(proj, proj_stride, ref, ref_dim, pad_mask), N, S, M, D, L, Lq, P, out,
                                      shapes_host, (hipStream_t)stream);
}

int msda_fused_forward_bf16(const uint16_t *value, const int64_t *shapes_dev, const int64_t *lstart_dev,
                            const float *proj, int proj_stride, const float *ref, int ref_dim, const uint8_t *pad_mask,
                            int N, int S, int M, int D, int L, int Lq, int P, uint16_t *out,
                            const int64_t *shapes_host, void *stream) {
    if (!proj) return fail(MSDA_EINVAL, "null pointer argument");
    return forward_impl<bf16_t, float>((const bf16_t *)value, shapes_dev, lstart_dev, nullptr, nullptr,
                                       fused_args(proj, proj_stride, ref, ref_dim, pad_mask), N, S, M, D, L, Lq, P,
                                       (bf16_t *)out, shapes_host, (hipStream_t)stream);
}

int msda_fused_backward_f32(const float *value, const int64_t *shapes_dev, const int64_t *lstart_dev,
                            const float *proj, int proj_stride, const float *ref, int ref_dim, const uint8_t *pad_mask,
                            const float *grad_out, int N, int S, int M, int D, int L, int Lq, int P, float *grad_value,
                            float *grad_proj, float *grad_ref_part, int zero_grad_value, const int64_t *shapes_host,
                            void *stream) {
    if (!proj) return fail(MSDA_EINVAL, "null pointer argument");
    return backward_impl<float, float, float>(value, shapes_dev, lstart_dev, nullptr, nullptr,
                                              fused_args(proj, proj_stride, ref, ref_dim, pad_mask), grad_out, N, S, M,
                                              D, L, Lq, P, grad_value, nullptr, nullptr, grad_proj, grad_ref_part,
                                              zero_grad_value, shapes_host, (hipStream_t)stream);
}

int msda_fused_backward_bf16(const uint16_t *value, const int64_t *shapes_dev, const int64_t *lstart_dev,
                             const float *proj, int proj_stride, const float *ref, int ref_dim, const uint8_t *pad_mask,
                             const uint16_t *grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                             float *grad_value, float *grad_proj, float *grad_ref_part, int zero_grad_value,
                             const int64_t *shapes_host, void *stream) {
    if (!proj) return fail(MSDA_EINVAL, "null pointer argument");
    return backward_impl<bf16_t, float, float>((const bf16_t *)value, shapes_dev, lstart_dev, nullptr, nullptr,
                                               fused_args(proj, proj_stride, ref, ref_dim, pad_mask),
                                               (const bf16_t *)grad_out, N, S, M, D, L, Lq, P, grad_value, nullptr,
                                               nullptr, grad_proj, grad_ref_part, zero_grad_value, shapes_host,
                                               (hipStream_t)stream);
}

// The fused backward given the forward's output of the same call (ABI 6): see backward_impl -- with it the default
// backward of the encoder's self-attention is ONE kernel (no attention-weight kernel in front, no Jacobian kernel behind).
int msda_fused_backward_out_f32(const float *value, const int64_t *shapes_dev, const int64_t *lstart_dev,
                                const float *proj, int proj_stride, const float *ref, int ref_dim,
                                const uint8_t *pad_mask, const float *grad_out, const float *fwd_out, int N, int S, int M,
                                int D, int L, int Lq, int P, float *grad_value, float *grad_proj, float *grad_ref_part,
                                int zero_grad_value, const int64_t *shapes_host, void *workspace, size_t workspace_bytes,
                                void *stream) {
    if (!proj) return fail(MSDA_EINVAL, "null pointer argument");
    return backward_impl<float, float, float>(value, shapes_dev, lstart_dev, nullptr, nullptr,
                                              fused_args(proj, proj_stride, ref, ref_dim, pad_mask), grad_out, N, S, M,
                                              D, L, Lq, P, grad_value, nullptr, nullptr, grad_proj, grad_ref_part,
                                              zero_grad_value, shapes_host, (hipStream_t)stream, (float *)workspace,
                                              workspace_bytes, fwd_out);
}

int msda_fused_backward_out_bf16(const uint16_t *value, const int64_t *shapes_dev, const int64_t *lstart_dev,
                                 const float *proj, int proj_stride, const float *ref, int ref_dim,
                                 const uint8_t *pad_mask, const uint16_t *grad_out, const uint16_t *fwd_out, int N, int S,
                                 int M, int D, int L, int Lq, int P, float *grad_value, float *grad_proj,
                                 float *grad_ref_part, int zero_grad_value, const int64_t *shapes_host, void *workspace,
                                 size_t workspace_bytes, void *stream) {
    if (!proj) return fail(MSDA_EINVAL, "null pointer argument");
    return backward_impl<bf16_t, float, float>((const bf16_t *)value, shapes_dev, lstart_dev, nullptr, nullptr,
                                               fused_args(proj, proj_stride, ref, ref_dim, pad_mask),
                                               (const bf16_t *)grad_out, N, S, M, D, L, Lq, P, grad_value, nullptr,
                                               nullptr, grad_proj, grad_ref_part, zero_grad_value, shapes_host,
                                               (hipStream_t)stream, (float *)workspace, workspace_bytes,
                                               (const bf16_t *)fwd_out);
}

size_t msda_fused_workspace_bytes(int N, int Lq, int M, int L, int P) {
    if (N < 0 || Lq < 0 || M <= 0 || L <= 0 || P <= 0) return 0;
    return (size_t)N * Lq * M * L * P * 3 * sizeof(float);
}

// Scratch the NEXT backward call of this thread's call site can use (ABI 6): the fused prologue's block
// (msda_fused_workspace_bytes) plus, when that call would build grad_value by sort + gather ("bwd_variant" 13, or
// selector level 2 for self-attention over the pyramid), the sort's records and tables.  0: the call needs none.
size_t msda_backward_workspace_bytes(int fused, int N, int S, int M, int D, int L, int Lq, int P, int elem_bytes,
                                     void *stream) {
    if (N <= 0 || S <= 0 || Lq <= 0 || M <= 0 || D <= 0 || L <= 0 || P <= 0) return 0;
    const size_t n_pts = (size_t)N * Lq * M * L * P;
    size_t bytes = fused ? n_pts * 3 * sizeof(float) : 0;
    const int variant = opt_bwd_variant.load();
    bool sorted = variant == 13;
    if (variant == 0 && opt_bwd_sorted.load() && D == 32 && Lq == S && L <= kTileMaxL && (elem_bytes == 4 || elem_bytes == 2))
        sorted = sel_peek(1, M, L, P, elem_bytes, (hipStream_t)stream) >= 2;
    if (sorted && D == 32) {
        SortPlan sp;
        if (make_sort_plan(sp, N, S, M, L, Lq, P, (size_t)elem_bytes, nullptr, opt_bwd_sort_qc.load(), opt_bwd_sort_emult.load())) bytes = ((bytes + 255) & ~(size_t)255) + sp.bytes;
    }
    return bytes;
}

int msda_backward_ws_f32(const float *value, const int64_t *shapes_dev, const int64_t *lstart_dev, const float *loc,
                         const float *attn, const float *grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                         float *grad_value, float *grad_loc, float *grad_attn, int zero_grad_value,
                         const int64_t *shapes_host, void *workspace, size_t workspace_bytes, void *stream) {
    return backward_impl<float, float, float>(value, shapes_dev, lstart_dev, loc, attn, FusedArgs{}, grad_out, N, S, M,
                                              D, L, Lq, P, grad_value, grad_loc, grad_attn, nullptr, nullptr,
                                              zero_grad_value, shapes_host, (hipStream_t)stream, (float *)workspace,
                                              workspace_bytes);
}

int msda_backward_ws_bf16(const uint16_t *value, const int64_t *shapes_dev, const int64_t *lstart_dev, const float *loc,
                          const float *attn, const uint16_t *grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                          float *grad_value, float *grad_loc, float *grad_attn, int zero_grad_value,
                          const int64_t *shapes_host, void *workspace, size_t workspace_bytes, void *stream) {
    return backward_impl<bf16_t, float, float>((const bf16_t *)value, shapes_dev, lstart_dev, loc, attn, FusedArgs{},
                                               (const bf16_t *)grad_out, N, S, M, D, L, Lq, P, grad_value, grad_loc,
                                               grad_attn, nullptr, nullptr, zero_grad_value, shapes_host,
                                               (hipStream_t)stream, (float *)workspace, workspace_bytes);
}

int msda_fused_backward_ws_f32(const float *value, const int64_t *shapes_dev, const int64_t *lstart_dev,
                               const float *proj, int proj_stride, const float *ref, int ref_dim,
                               const uint8_t *pad_mask, const float *grad_out, int N, int S, int M, int D, int L, int Lq,
                               int P, float *grad_value, float *grad_proj, float *grad_ref_part, int zero_grad_value,
                               const int64_t *shapes_host, void *workspace, size_t workspace_bytes, void *stream) {
    if (!proj) return fail(MSDA_EINVAL, "null pointer argument");
    return backward_impl<float, float, float>(value, shapes_dev, lstart_dev, nullptr, nullptr,
                                              fused_args(proj, proj_stride, ref, ref_dim, pad_mask), grad_out, N, S, M,
                                              D, L, Lq, P, grad_value, nullptr, nullptr, grad_proj, grad_ref_part,
                                              zero_grad_value, shapes_host, (hipStream_t)stream, (float *)workspace,
                                              workspace_bytes);
}

int msda_fused_backward_ws_bf16(const uint16_t *value, const int64_t *shapes_dev, const int64_t *lstart_dev,
                                const float *proj, int proj_stride, const float *ref, int ref_dim,
                                const uint8_t *pad_mask, const uint16_t *grad_out, int N, int S, int M, int D, int L,
                                int Lq, int P, float *grad_value, float *grad_proj, float *grad_ref_part,
                                int zero_grad_value, const int64_t *shapes_host, void *workspace,
                                size_t workspace_bytes, void *stream) {
    if (!proj) return fail(MSDA_EINVAL, "null pointer argument");
    return backward_impl<bf16_t, float, float>((const bf16_t *)value, shapes_dev, lstart_dev, nullptr, nullptr,
                                               fused_args(proj, proj_stride, ref, ref_dim, pad_mask),
                                               (const bf16_t *)grad_out, N, S, M, D, L, Lq, P, grad_value, nullptr,
                                               nullptr, grad_proj, grad_ref_part, zero_grad_value, shapes_host,
                                               (hipStream_t)stream, (float *)workspace, workspace_bytes);
}

int msda_fused_points_f32(const int64_t *shapes_dev, const float *proj, int proj_stride, const float *ref, int ref_dim,
                          int N, int M, int L, int Lq, int P, float *loc_out, float *attn_out, void *stream) {
    if (!shapes_dev || !proj || !ref || !loc_out || !attn_out) return fail(MSDA_EINVAL, "null pointer argument");
    if (N < 0 || Lq < 0 || M <= 0 || L <= 0 || P <= 0) return fail(MSDA_EINVAL, "bad dimension");
    const FusedArgs fa = fused_args(proj, proj_stride, ref, ref_dim, nullptr);
    const int rc = check_fused(fa, M, L, P);
    if (rc) return rc;
    const long n_rows = (long)N * Lq * M;
    if (n_rows == 0) { g_err[0] = 0; return MSDA_OK; }
    const PointSrc src = make_src(nullptr, nullptr, fa, M, L, P);
    if (L * P <= 16) {
        hipLaunchKernelGGL(msda_fused_points16_kernel, dim3(clamp_grid((n_rows * 16 + 255) / 256, 32)), dim3(256), 0,
                           (hipStream_t)stream, shapes_dev, src, n_rows, M, L, P, loc_out, attn_out);
        return check_launch("msda_fused_points16_kernel");
    }
    const int grid = clamp_grid((n_rows * 8 + 255) / 256, 16);
    hipLaunchKernelGGL(msda_fused_points_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, shapes_dev, src, n_rows,
                       M, L, P, loc_out, attn_out);
    return check_launch("msda_fused_points_kernel");
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
    if (!strcmp(key, "bwd_tile_margin")) return &opt_bwd_tile_margin;
    if (!strcmp(key, "bwd_ablate")) return &opt_bwd_ablate;
    if (!strcmp(key, "bwd_wide_log2")) return &opt_bwd_wide_log2;
    if (!strcmp(key, "bwd_split")) return &opt_bwd_split;
    if (!strcmp(key, "bwd_bins_margin")) return &opt_bwd_bins_margin;
    if (!strcmp(key, "bwd_bins_margin_hi")) return &opt_bwd_bins_margin_hi;
    if (!strcmp(key, "auto_select")) return &opt_auto_select;
    if (!strcmp(key, "deterministic")) return &opt_deterministic;
    if (!strcmp(key, "sel_level")) return &opt_sel_level;
    if (!strcmp(key, "sel_up0")) return &opt_sel_up0;
    if (!strcmp(key, "sel_up1")) return &opt_sel_up1;
    if (!strcmp(key, "sel_up1_rows")) return &opt_sel_up1_rows;
    if (!strcmp(key, "sel_down2_rows")) return &opt_sel_down2_rows;
    if (!strcmp(key, "sel_down1")) return &opt_sel_down1;
    if (!strcmp(key, "sel_down2")) return &opt_sel_down2;
    if (!strcmp(key, "sel_fwd_up")) return &opt_sel_fwd_up;
    if (!strcmp(key, "sel_fwd_down")) return &opt_sel_fwd_down;
    if (!strcmp(key, "bwd_bins_strip")) return &opt_bwd_bins_strip;
    if (!strcmp(key, "bwd_rows")) return &opt_bwd_rows;
    if (!strcmp(key, "bwd_sorted")) return &opt_bwd_sorted;
    if (!strcmp(key, "bwd_soft")) return &opt_bwd_soft;
    if (!strcmp(key, "bwd_sort_qc")) return &opt_bwd_sort_qc;
    if (!strcmp(key, "bwd_sort_emult")) return &opt_bwd_sort_emult;
    if (!strcmp(key, "bwd_rows_block")) return &opt_bwd_rows_block;
    if (!strcmp(key, "fwd_win_rlog")) return &opt_fwd_win_rlog;
    if (!strcmp(key, "fwd_win_rlogx")) return &opt_fwd_win_rlogx;
    if (!strcmp(key, "fwd_win_auto")) return &opt_fwd_win_auto;
    if (!strcmp(key, "fwd_win_bf16")) return &opt_fwd_win_bf16;
    if (!strcmp(key, "fwd_head_major")) return &opt_fwd_head_major;
    if (!strcmp(key, "fwd_win_block")) return &opt_fwd_win_block;
    if (!strcmp(key, "fwd_win_l0")) return &opt_fwd_win_l0;
    if (!strcmp(key, "fwd_win_margins")) return &opt_fwd_win_margins;
    if (!strcmp(key, "bwd_side_rows")) return &opt_bwd_side_rows;
    if (!strcmp(key, "fwd_win_trace_lo")) return &opt_fwd_win_trace_lo;
    if (!strcmp(key, "fwd_win_trace_hi")) return &opt_fwd_win_trace_hi;
    if (!strcmp(key, "fwd_win_ablate")) return &opt_fwd_win_ablate;
    if (!strcmp(key, "fwd_win_place")) return &opt_fwd_win_place;
    if (!strcmp(key, "fwd_win_wps")) return &opt_fwd_win_wps;
    if (!strcmp(key, "fwd_win_grid")) return &opt_fwd_win_grid;
    if (!strcmp(key, "fwd_win_rsy")) return &opt_fwd_win_rsy;
    if (!strcmp(key, "fwd_win_rsx")) return &opt_fwd_win_rsx;
    if (!strcmp(key, "fwd_win_early")) return &opt_fwd_win_early;
    return nullptr;
}

int msda_set_option(const char *key, int value) {
    std::atomic<int> *o = find_opt(key);
    if (!o || (value < 0 && !(o == &opt_sel_level && value == -1)))
        return fail(MSDA_EINVAL, "unknown option or negative value");
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
