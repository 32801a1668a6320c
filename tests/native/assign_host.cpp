// Host build of memotr_amd/csrc/assign_core.h for tests/test_assign_core.py: the assignment solver the device kernel
// (clip_ops.hip, clipops_assign_f32) is instantiated from, with 1 or 64 virtual lanes (the latter walks the exact
// partition and cross-lane reduction of the wavefront).  Built by the test with g++; not part of the product.
#include <stdlib.h>

#include "../../memotr_amd/csrc/assign_core.h"

extern "C" int assign_host(const float *cost, int n_rows, int n_cols, int lanes, int32_t *row_ind, int32_t *col_ind) {
    const int nr = n_rows < n_cols ? n_rows : n_cols, nc = n_rows < n_cols ? n_cols : n_rows;
    void *mem = malloc(assign::work_bytes(nr, nc) + 64);
    int rc;
    if (lanes == 64)
        rc = assign::solve_problem(assign::SerialLanes<64>{}, cost, n_cols, 1, n_rows, n_cols, mem, row_ind, col_ind);
    else if (lanes == 7)
        rc = assign::solve_problem(assign::SerialLanes<7>{}, cost, n_cols, 1, n_rows, n_cols, mem, row_ind, col_ind);
    else
        rc = assign::solve_problem(assign::SerialLanes<1>{}, cost, n_cols, 1, n_rows, n_cols, mem, row_ind, col_ind);
    free(mem);
    return rc;
}
