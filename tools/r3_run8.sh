#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r3_run8_tests.log 2>&1
tail -4 gpurun_out/r3_run8_tests.log
timeout 600 python tools/kbench.py --quick --dists encoder_like --out gpurun_out/r3_kbench8.json > gpurun_out/r3_kbench8.log 2>&1
grep "dec320" gpurun_out/r3_kbench8.log | grep bwd
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3_bench8.json 2> gpurun_out/r3_bench8.err; cut -c1-200 gpurun_out/r3_bench8.json
bash tools/prof.sh r03a train > gpurun_out/r3_prof_train.log 2>&1
ls gpurun_out/profiles | tail -5
