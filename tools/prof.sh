#!/bin/bash
# rocprofv3 passes for the bench workload (run on the GPU box, from the repo root).
#   tools/prof.sh <tag> [train|msda]  -> gpurun_out/prof_<tag>/{stats,pmc_fetch,pmc_write}/...
# Kernel-trace/stats and the two PMC passes are separate runs (FETCH_SIZE and WRITE_SIZE do not fit
# one pass; never combined with sys/hip/hsa tracing).
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
WORKLOAD=${2:-train}
# PROF_ARGS: extra bench.py flags, e.g. "--dtype bf16", "--config mot17 --use-checkpoint", "--config bdd100k --dtype bf16"
CMD="python bench.py --workload $WORKLOAD --steps 3 --warmup 2 --no-cpu-baseline ${PROF_ARGS:-}"
# MIOpen benchmarks every applicable solver (its naive reference kernels included) the first time a process on this
# box meets a convolution shape and stores the pick in ~/.config/miopen (MIOPEN_FIND_MODE DYNAMIC_HYBRID).  On a fresh
# box the profiled process would be that first process and its kernel table would be the find phase, not the train
# step: prime the user find-db with one un-profiled run first.
# the train-step table holds the step only: bench.py's own kernel-timing legs (~1500 launches of the encoder call) off
if [ "$WORKLOAD" = "train" ]; then export MEMOTR_BENCH_NO_KERNEL_LEGS=1; $CMD > "$OUT/prime.log" 2>&1; fi
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- $CMD > "$OUT/stats.log" 2>&1
if [ "$WORKLOAD" = "train" ]; then
  # counter passes serialise ~30k dispatches per step: only the stats pass for the full train step
  python tools/prof_summary.py "$OUT" "$TAG" "$CMD"
  mkdir -p gpurun_out/profiles && cp "$OUT"/stats/*kernel_stats.csv gpurun_out/profiles/${TAG}_kernel_stats.csv 2>/dev/null
  rm -rf "$OUT"; exit 0
fi
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o fetch -- $CMD > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_write" -o write -- $CMD > "$OUT/pmc_write.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT/pmc_l2" -o l2 -- $CMD > "$OUT/pmc_l2.log" 2>&1
find "$OUT" -name '*.csv' | head -50
python tools/prof_summary.py "$OUT" "$TAG" "$CMD"
# keep the stats table, drop the per-dispatch traces (tens of MB)
mkdir -p gpurun_out/profiles && cp "$OUT"/stats/*kernel_stats.csv gpurun_out/profiles/${TAG}_kernel_stats.csv 2>/dev/null
rm -rf "$OUT"
