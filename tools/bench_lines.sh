#!/bin/bash
# the bench lines kept under profiles/ per round (run on the GPU box from the repo root):  tools/bench_lines.sh [bdd|rest|all] [r04]
# (REFIND=1: MIOpen's exhaustive find into a scratch db -- ~9 GPU-minutes for the BDD100K pyramid; the default uses the
#  find-db that travels with the package, memotr_amd/tuning/miopen_db, which bench.py points MIOpen at)
R=${2:-r06}
mkdir -p gpurun_out/lines gpurun_out/miopen/db gpurun_out/miopen/cache
if [ "${REFIND:-0}" = "1" ]; then
  export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/miopen/db
  export MIOPEN_CUSTOM_CACHE_DIR=$PWD/gpurun_out/miopen/cache
fi
run() { name=$1; shift; t0=$SECONDS; python bench.py "$@" 2> gpurun_out/lines/$name.err | tail -1 > gpurun_out/lines/$name.json; echo "$name: $((SECONDS - t0)) s wall"; cut -c1-160 gpurun_out/lines/$name.json; }
case "${1:-all}" in
  bdd|all)
    run ${R}_bench_config5_bdd100k_bf16 --config bdd100k --dtype bf16 --steps 10 --warmup 14 --no-cpu-baseline
    run ${R}_bench_config5_bdd100k_f32 --config bdd100k --dtype f32 --steps 10 --warmup 14 --no-cpu-baseline
    du -sh gpurun_out/miopen/db gpurun_out/miopen/cache ;;&
  rest|all)
    run ${R}_bench_config4_mot17_checkpoint --config mot17 --use-checkpoint --steps 8 --warmup 3 --no-cpu-baseline
    run ${R}_bench_train_bf16 --dtype bf16 --steps 10 --warmup 4 --no-cpu-baseline
    run ${R}_bench_train --steps 10 --warmup 4
    run ${R}_bench_infer --workload infer --no-cpu-baseline
    run ${R}_bench_msda --workload msda --no-cpu-baseline ;;
esac
