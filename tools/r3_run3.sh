#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_msda_fwd_win_gpu.py -q -m gpu -x > gpurun_out/r3_run3_tests.log 2>&1
tail -5 gpurun_out/r3_run3_tests.log
timeout 1200 python tools/kbench.py --fwd-only --dists encoder_like --out gpurun_out/r3_kbench3.json > gpurun_out/r3_kbench3.log 2>&1
grep "v12\|v3 " gpurun_out/r3_kbench3.log | grep -v dec320
timeout 400 tools/pmc_probe.sh r3_win_v2 fwd fwd_variant=12 > /dev/null 2>&1
timeout 400 tools/pmc_probe.sh r3_win_v2_plain fwd plain fwd_variant=12 > /dev/null 2>&1
cat gpurun_out/pmc_r3_win_v2.txt gpurun_out/pmc_r3_win_v2_plain.txt
