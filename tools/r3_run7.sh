#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_clip_ops_gpu.py -q -m gpu > gpurun_out/r3_run7_tests.log 2>&1
tail -4 gpurun_out/r3_run7_tests.log
python bench.py --config mot17 --use-checkpoint --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3_bench_config4.json 2> gpurun_out/r3_c4.err; cut -c1-330 gpurun_out/r3_bench_config4.json; grep -o '"decoder_graph_stats": {[^}]*}' gpurun_out/r3_bench_config4.json
timeout 1500 python bench.py --config bdd100k --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3_bench_config5_bf16.json 2> gpurun_out/r3_c5.err; cut -c1-330 gpurun_out/r3_bench_config5_bf16.json; tail -2 gpurun_out/r3_c5.err
timeout 900 python bench.py --config bdd100k --dtype f32 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3_bench_config5_f32.json 2> gpurun_out/r3_c5f.err; cut -c1-330 gpurun_out/r3_bench_config5_f32.json
