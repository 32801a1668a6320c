#!/usr/bin/env python
"""Dense / sparse segments of a step sequence written by tools/small_trace.py (argument 3): runs of kernels that follow
each other within 6 us (a replayed graph, or a GPU-bound stretch) and the host-bound stretches between them."""
import sys

rows = []
for line in open(sys.argv[1]):
    p = line.split(None, 3)
    rows.append((float(p[0]), float(p[1]), int(p[2]), p[3].strip()))
MIN = int(sys.argv[2]) if len(sys.argv) > 2 else 30
segs, cur = [], [0]
for i in range(1, len(rows)):
    if rows[i][0] - (rows[i - 1][0] + rows[i - 1][1]) < 6:
        cur.append(i)
    else:
        segs.append(cur)
        cur = [i]
segs.append(cur)
prev_end, sparse_n, sparse_busy = 0.0, 0, 0.0
for s in segs:
    a, b = s[0], s[-1]
    if len(s) >= MIN:
        t0, t1 = rows[a][0], rows[b][0] + rows[b][1]
        if sparse_n:
            print(f"   host-bound {sparse_n:4d} kernels  {prev_end / 1e3:8.3f} -> {t0 / 1e3:8.3f} ms  ({(t0 - prev_end) / 1e3:.3f} ms, "
                  f"{sparse_busy / max(t0 - prev_end, 1e-9) * 100:.0f} % busy)")
        print(f"dense {len(s):4d} kernels  {t0 / 1e3:8.3f} -> {t1 / 1e3:8.3f} ms  ({(t1 - t0) / 1e3:.3f} ms) first {rows[a][3][:44]}")
        prev_end, sparse_n, sparse_busy = t1, 0, 0.0
    else:
        sparse_n += len(s)
        sparse_busy += sum(rows[i][1] for i in s)
