#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_msda_fwd_win_gpu.py -q -m gpu > gpurun_out/r3_run2_tests.log 2>&1
tail -15 gpurun_out/r3_run2_tests.log
timeout 200 tools/ubench/ubench > gpurun_out/r3_ubench.log 2>&1
grep -i "pair\|ideal\|lds_read" gpurun_out/r3_ubench.log
timeout 1200 python tools/kbench.py --fwd-only --dists encoder_like --out gpurun_out/r3_kbench2.json > gpurun_out/r3_kbench2.log 2>&1
grep "v12\|v3 " gpurun_out/r3_kbench2.log | grep -v dec320
