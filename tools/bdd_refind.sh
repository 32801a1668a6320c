#!/bin/bash
# BDD100K fp32 line with (a) the find-db that travels with the package, (b) MIOpen's exhaustive find into a scratch db
# (minutes), (c) the scratch db again (answers from the db now)
export MEMOTR_BENCH_NO_KERNEL_LEGS=1
run() { name=$1; shift; t0=$SECONDS; env "$@" python bench.py --config bdd100k --dtype f32 --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$name', round(d['ms_per_step'],2), round(d['value'],2))"; echo "  ($((SECONDS - t0)) s wall)"; }
run shipped-db A=1
mkdir -p gpurun_out/miopen/db gpurun_out/miopen/cache
run refind MIOPEN_USER_DB_PATH=$PWD/gpurun_out/miopen/db MIOPEN_CUSTOM_CACHE_DIR=$PWD/gpurun_out/miopen/cache
run refind-again MIOPEN_USER_DB_PATH=$PWD/gpurun_out/miopen/db MIOPEN_CUSTOM_CACHE_DIR=$PWD/gpurun_out/miopen/cache
ls -la gpurun_out/miopen/db
python -c "import torch; print(torch.__version__, torch.backends.cudnn.version())"
