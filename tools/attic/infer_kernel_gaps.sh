#!/bin/bash
# Is the GPU idle or busy while an online-tracking frame stalls?  rocprofv3 kernel trace of bench.py --workload infer,
# then the largest gaps between consecutive kernels on the device (run on the GPU box from the repo root).
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/infer_trace; rm -rf $OUT; mkdir -p $OUT
# (prime MIOpen's find-db first: on a fresh box the profiled process would otherwise be the find phase, see prof.sh)
python bench.py --workload infer --no-cpu-baseline --steps 5 > $OUT/prime.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python bench.py --workload infer --no-cpu-baseline --steps 40 > $OUT/run.log 2>&1
tail -1 $OUT/run.log | cut -c1-140
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/infer_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in csv.DictReader(open(f))))
print("kernels", len(rows), "span %.1f ms" % ((rows[-1][1] - rows[0][0]) / 1e6))
end = rows[0][1]
gaps = []
for (s, e, n), prev in zip(rows[1:], rows[:-1]):
    if s - end > 0:
        gaps.append((s - end, prev[2], n, (end - rows[0][0]) / 1e6))
    end = max(end, e)
gaps.sort(reverse=True)
print("idle gaps > 20 ms:", sum(1 for g in gaps if g[0] > 20e6), " total idle %.1f ms" % (sum(g[0] for g in gaps) / 1e6))
for g in gaps[:12]:
    print("  %6.1f ms idle at t=%8.1f ms after %-60s before %s" % (g[0] / 1e6, g[3], g[1], g[2]))
PY
rm -rf $OUT
