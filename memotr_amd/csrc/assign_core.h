// assign_core.h -- rectangular linear sum assignment, bit-for-bit the decisions of scipy.optimize.linear_sum_assignment.
//
// What it replaces: the host call of the reference's matcher (models/matcher.py:122-124,
// `linear_sum_assignment(c[i])` on a `.cpu()` copy of the cost matrix).  scipy solves the problem with the shortest
// augmenting path form of the Jonker-Volgenant algorithm (D. F. Crouse, "On implementing 2D rectangular assignment
// algorithms", IEEE TAES 52(4), 2016) in double precision; this header restates that algorithm -- the scan order of
// the unvisited columns (a list kept in reverse, removal by swapping with the last entry), the tie rule of the scan
// (a strictly smaller reduced cost wins; an equal one wins only if its column is still unassigned), the dual updates
// and the augmentation -- so that ties resolve the same way, and the tests hold it to scipy's output on random
// matrices with and without ties (tests/test_assign_core.py on the CPU, tests/test_clip_ops_gpu.py on the GPU).
//
// The same source compiles for the host (g++: the CPU test harness, tests/native/assign_host.cpp) and for the
// device (hipcc: clip_ops.hip, one 64-lane wavefront per problem).  `Lanes` abstracts the only data-parallel step,
// the scan over the unvisited columns:
//   * host: `SerialLanes<W>` walks the W virtual lanes one after the other (W = 1 is the textbook loop; W = 64
//     exercises exactly the partition and the cross-lane reduction the device uses);
//   * device: `WaveLanes` (clip_ops.hip) runs them on the wavefront and reduces with DPP / readlane.
// Everything else is executed by lane 0 / redundantly uniform.
#pragma once

#include <stdint.h>

#ifdef __HIPCC__
#define ASSIGN_HD __host__ __device__ __forceinline__
#else
#define ASSIGN_HD inline
#endif

namespace assign {

constexpr double kInf = __builtin_inf();

// scratch of one problem with nr <= nc (after the optional transposition); all arrays live in LDS on the device
struct Work {
    double *u;                  // [nr] dual variables of the rows
    double *v;                  // [nc] dual variables of the columns
    double *shortest;           // [nc] shortest path cost to column j
    int32_t *path;              // [nc] predecessor row of column j
    int32_t *col4row;           // [nr]
    int32_t *row4col;           // [nc]
    int32_t *remaining;         // [nc] unvisited columns, in scipy's order
    uint8_t *SR;                // [nr] rows on the alternating tree
    uint8_t *SC;                // [nc] columns on the alternating tree
};

ASSIGN_HD size_t work_bytes(int nr, int nc) {
    // doubles first (8-byte aligned), then ints, then bytes
    return (size_t)(nr + 2 * nc) * 8 + (size_t)(nr + 3 * nc) * 4 + (size_t)(nr + nc);
}

ASSIGN_HD Work carve(void *mem, int nr, int nc) {
    Work w;
    double *d = (double *)mem;
    w.u = d;
    w.v = d + nr;
    w.shortest = d + nr + nc;
    int32_t *i = (int32_t *)(d + nr + 2 * nc);
    w.path = i;
    w.col4row = i + nc;
    w.row4col = i + nc + nr;
    w.remaining = i + 2 * nc + nr;
    uint8_t *b = (uint8_t *)(i + 3 * nc + nr);
    w.SR = b;
    w.SC = b + nr;
    return w;
}

// cost(i, j) of the (possibly transposed) problem, widened to double exactly as scipy's float64 conversion does
struct CostView {
    const float *base;
    long stride_i, stride_j;    // in elements
    ASSIGN_HD double at(int i, int j) const { return (double)base[(long)i * stride_i + (long)j * stride_j]; }
};

// One lane's share of a scan: the minimum of the entries it looked at, the LAST position among them whose column is
// unassigned, and the FIRST position among them (sequential rule: "< lowest, or == lowest and unassigned", walked in
// ascending position -- the final choice is the last unassigned position holding the global minimum if there is one,
// else the first position holding it).
struct ScanBest {
    double lowest;
    int32_t last_free;          // -1: none
    int32_t first;              // -1: nothing scanned
};

ASSIGN_HD ScanBest scan_empty() { return ScanBest{kInf, -1, -1}; }

ASSIGN_HD void scan_take(ScanBest &b, double val, int32_t it, bool unassigned) {
    if (val < b.lowest) {
        b.lowest = val;
        b.first = it;
        b.last_free = unassigned ? it : -1;
    } else if (val == b.lowest) {
        if (b.first < 0) b.first = it;                  // (val == inf on an empty record)
        if (unassigned) b.last_free = it;
    }
}

ASSIGN_HD ScanBest scan_merge(const ScanBest &a, const ScanBest &b) {
    if (a.first < 0) return b;
    if (b.first < 0) return a;
    if (a.lowest < b.lowest) return a;
    if (b.lowest < a.lowest) return b;
    ScanBest o;
    o.lowest = a.lowest;
    o.first = a.first < b.first ? a.first : b.first;
    o.last_free = a.last_free > b.last_free ? a.last_free : b.last_free;
    return o;
}

ASSIGN_HD int32_t scan_choice(const ScanBest &b) { return b.last_free >= 0 ? b.last_free : b.first; }

// Host stand-in for the wavefront: W virtual lanes, lane l takes positions l, l + W, l + 2W, ...
template <int W>
struct SerialLanes {
    static constexpr int width = W;
    template <typename F>
    ASSIGN_HD ScanBest scan(int n, F &&body) const {       // body(it) -> visits position it, returns (val, unassigned)
        ScanBest total = scan_empty();
        for (int lane = 0; lane < W; ++lane) {
            ScanBest mine = scan_empty();
            for (int it = lane; it < n; it += W) body(it, mine);
            total = scan_merge(total, mine);
        }
        return total;
    }
    template <typename F>
    ASSIGN_HD void each(int n, F &&body) const {
        for (int k = 0; k < n; ++k) body(k);
    }
    ASSIGN_HD bool leader() const { return true; }
    ASSIGN_HD void sync() const {}
};

// Solve one problem with nr <= nc.  Returns false when no complete assignment exists (an infinite column set).
template <typename Lanes>
ASSIGN_HD bool solve_rows_le_cols(const Lanes &lanes, const CostView &cost, int nr, int nc, const Work &w) {
    lanes.each(nr, [&](int i) { w.u[i] = 0.0; w.col4row[i] = -1; });
    lanes.each(nc, [&](int j) { w.v[j] = 0.0; w.row4col[j] = -1; w.path[j] = -1; });
    lanes.sync();
    for (int cur = 0; cur < nr; ++cur) {
        // ---- shortest augmenting path from row `cur` ----
        lanes.each(nc, [&](int it) { w.remaining[it] = nc - it - 1; w.shortest[it] = kInf; w.SC[it] = 0; });
        lanes.each(nr, [&](int i) { w.SR[i] = 0; });
        lanes.sync();
        double min_val = 0.0;
        int num_remaining = nc;
        int sink = -1;
        int i = cur;
        while (sink == -1) {
            if (lanes.leader()) w.SR[i] = 1;
            const double ui = w.u[i];
            const ScanBest best = lanes.scan(num_remaining, [&](int it, ScanBest &mine) {
                const int j = w.remaining[it];
                const double r = min_val + cost.at(i, j) - ui - w.v[j];
                double s = w.shortest[j];
                if (r < s) {
                    w.path[j] = i;
                    w.shortest[j] = r;
                    s = r;
                }
                scan_take(mine, s, it, w.row4col[j] == -1);
            });
            min_val = best.lowest;
            if (!(min_val < kInf)) return false;          // infeasible cost matrix
            const int index = scan_choice(best);
            lanes.sync();
            const int j = w.remaining[index];
            const int owner = w.row4col[j];
            if (owner == -1) sink = j;
            else i = owner;
            lanes.sync();
            if (lanes.leader()) {
                w.SC[j] = 1;
                w.remaining[index] = w.remaining[--num_remaining];
            } else {
                --num_remaining;
            }
            lanes.sync();
        }
        // ---- dual variables ----
        lanes.each(nr, [&](int r) {
            if (r == cur) w.u[r] += min_val;
            else if (w.SR[r]) w.u[r] += min_val - w.shortest[w.col4row[r]];
        });
        lanes.each(nc, [&](int j) {
            if (w.SC[j]) w.v[j] -= min_val - w.shortest[j];
        });
        lanes.sync();
        // ---- augment ----
        if (lanes.leader()) {
            int j = sink;
            while (true) {
                const int r = w.path[j];
                w.row4col[j] = r;
                const int prev = w.col4row[r];
                w.col4row[r] = j;
                j = prev;
                if (r == cur) break;
            }
        }
        lanes.sync();
    }
    return true;
}

// One (n_rows x n_cols) problem, cost row-major with the given strides: min(n_rows, n_cols) pairs, ordered by row
// index like scipy's result (`row_ind` ascending).  `mem` holds work_bytes(min, max) bytes.  Returns the number of
// pairs, or -1 for an infeasible matrix.
template <typename Lanes>
ASSIGN_HD int solve_problem(const Lanes &lanes, const float *cost, long stride_row, long stride_col, int n_rows,
                            int n_cols, void *mem, int32_t *row_ind, int32_t *col_ind) {
    if (n_rows <= 0 || n_cols <= 0) return 0;
    const bool transposed = n_cols < n_rows;             // scipy: "tall" problems are solved on the transpose
    const int nr = transposed ? n_cols : n_rows, nc = transposed ? n_rows : n_cols;
    const CostView view{cost, transposed ? stride_col : stride_row, transposed ? stride_row : stride_col};
    const Work w = carve(mem, nr, nc);
    if (!solve_rows_le_cols(lanes, view, nr, nc, w)) return -1;
    if (lanes.leader()) {
        if (!transposed) {
            for (int i = 0; i < nr; ++i) {
                row_ind[i] = i;
                col_ind[i] = w.col4row[i];
            }
        } else {                                          // internal column j = original row j
            int k = 0;
            for (int j = 0; j < nc; ++j)
                if (w.row4col[j] != -1) {
                    row_ind[k] = j;
                    col_ind[k] = w.row4col[j];
                    ++k;
                }
        }
    }
    lanes.sync();
    return nr;
}

}  // namespace assign
