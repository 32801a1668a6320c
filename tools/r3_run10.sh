#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python bench.py --dtype bf16 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r3_bench_bf16_dt.json 2> gpurun_out/r3_bench_bf16_dt.err; cut -c1-200 gpurun_out/r3_bench_bf16_dt.json
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bf16 -o stats -- python $GRAFT_REPO_ROOT/bench.py --dtype bf16 --steps 3 --warmup 2 --no-cpu-baseline > /tmp/prof_bf16.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/prof_bf16/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6, "kernels", sum(int(r["Calls"]) for r in rows))
for r in rows[:60]:
    print(f'{r["Name"][:100]:100s} {int(r["Calls"]):6d} {float(r["TotalDurationNs"])/1e6:9.2f} ms {float(r["AverageNs"])/1e3:9.1f} us')
PY
