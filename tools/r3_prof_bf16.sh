#!/bin/bash
# kernel-time table of the bf16 train step (top kernels + total per step); args: extra bench flags
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_bf16; rm -rf $OUT; mkdir -p $OUT
CMD="python bench.py --dtype bf16 --steps 3 --warmup 2 --no-cpu-baseline $*"
$CMD > $OUT/prime.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $CMD > $OUT/stats.log 2>&1
tail -1 $OUT/stats.log | cut -c1-200
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_bf16/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms over 5 steps: %.1f  -> per step %.1f ms" % (tot / 1e6, tot / 5e6))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:32]:
    print("%8.2f ms/step %6d calls/step  %s" % (float(r["TotalDurationNs"]) / 5e6, int(r["Calls"]) // 5, r["Name"][:130]))
PY
mkdir -p gpurun_out/profiles; cp $OUT/stats/*/*kernel_stats.csv gpurun_out/profiles/r03_bf16_train_kernel_stats.csv 2>/dev/null || cp $(find $OUT -name '*kernel_stats.csv' | head -1) gpurun_out/profiles/r03_bf16_train_kernel_stats.csv
rm -rf $OUT
