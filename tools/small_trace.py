#!/usr/bin/env python
"""Round 6: the small launches of ONE train step from a rocprofv3 kernel trace (graphs ON, so torch.profiler cannot
attribute them): per kernel family (add / fill / copy / mul / cat ...) a histogram over grid sizes and over the kernel
that ran just before -- which names the neighbourhood (decoder graph, criterion, query updater, optimizer ...).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr -o tr -- python bench.py --workload train --steps 2 --warmup 2 --no-cpu-baseline
    python tools/small_trace.py gpurun_out/tr/*/tr_kernel_trace.csv > gpurun_out/small_trace.txt
"""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", name)
    m = re.match(r"vectorized_elementwise_kernel<\d+, (\w+(?:<[\w, ]+>)?)", name)
    if m:
        return "vec:" + m.group(1)
    m = re.match(r"elementwise_kernel_manual_unroll<\d+, \d+, gpu_kernel_impl(?:_nocast)?<(.{0,60})", name)
    if m:
        return "ew:" + m.group(1)
    return re.sub(r"\(.*", "", name)[:70]


rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]),
                     int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)))
rows.sort()
# step boundaries: the optimizer's multi_tensor_apply bursts (a gap of > 20 ms between bursts = one step)
opt = [i for i, r in enumerate(rows) if r[2].startswith("multi_tensor_apply")]
bounds = [opt[0]] + [b for a, b in zip(opt, opt[1:]) if rows[b][0] - rows[a][0] > 20_000_000]
if len(bounds) < 2:
    sys.exit("fewer than two steps in the trace")
lo, hi = bounds[-2], bounds[-1]
step = rows[lo:hi]
wall = (step[-1][1] - step[0][0]) / 1e6
busy = sum(e - s for s, e, _, _ in step) / 1e6
print(f"# last step of the trace: {len(step)} launches, {wall:.2f} ms first start -> last end, {busy:.2f} ms of kernel time")
small = [r for r in step if r[1] - r[0] < 20_000]
print(f"# launches under 20 us: {len(small)}, {sum(e - s for s, e, _, _ in small) / 1e6:.2f} ms")
fam = collections.defaultdict(lambda: [0, 0])
for s, e, n, g in step:
    fam[n][0] += 1
    fam[n][1] += e - s
print("\n# families by launches")
for n, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"{c:6d} {t / 1e6:8.2f} ms  {n}")
WATCH = sys.argv[2].split(",") if len(sys.argv) > 2 else ["vec:CUDAFunctor_add", "vec:FillFunctor", "__amd_rocclr_copyBuffer",
                                                          "vec:BinaryFunctor", "CatArrayBatchedCopy", "reduce_kernel"]
for w in WATCH:
    by_grid = collections.defaultdict(lambda: [0, 0])
    by_prev = collections.defaultdict(lambda: [0, 0])
    by_next = collections.defaultdict(lambda: [0, 0])
    for i, (s, e, n, g) in enumerate(step):
        if not n.startswith(w):
            continue
        by_grid[g][0] += 1
        by_grid[g][1] += e - s
        p = step[i - 1][2] if i else "-"
        by_prev[p][0] += 1
        by_prev[p][1] += e - s
        q = step[i + 1][2] if i + 1 < len(step) else "-"
        by_next[q][0] += 1
    print(f"\n## {w}: by grid size (threads)")
    for g, (c, t) in sorted(by_grid.items(), key=lambda kv: -kv[1][0])[:25]:
        print(f"{c:6d} {t / 1e6:8.2f} ms  grid {g}")
    print(f"## {w}: by the kernel before")
    for p, (c, t) in sorted(by_prev.items(), key=lambda kv: -kv[1][0])[:25]:
        print(f"{c:6d} {t / 1e6:8.2f} ms  after {p}")
    print(f"## {w}: by the kernel after")
    for p, (c, t) in sorted(by_next.items(), key=lambda kv: -kv[1][0])[:15]:
        print(f"{c:6d}  before {p}")
if len(sys.argv) > 3:       # the step as a sequence: dur us | grid | name
    with open(sys.argv[3], "w") as f:
        for s, e, n, g in step:
            f.write(f"{(s - step[0][0]) / 1e3:10.1f} {(e - s) / 1e3:8.2f} {g:9d} {n}\n")
