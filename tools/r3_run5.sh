#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r3_run5_tests.log 2>&1
tail -6 gpurun_out/r3_run5_tests.log
python bench.py --workload infer --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r3_bench_infer.json 2> gpurun_out/r3_bench_infer.err; tail -3 gpurun_out/r3_bench_infer.err; cat gpurun_out/r3_bench_infer.json
