#!/bin/bash
# round-3 GPU call 1: parity of the windowed forward, LDS pair-read ceiling, forward sweep, counters
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_msda_fwd_win_gpu.py -q -m gpu > gpurun_out/r3_run1_tests.log 2>&1
tail -15 gpurun_out/r3_run1_tests.log
timeout 200 tools/ubench/ubench > gpurun_out/r3_ubench.log 2>&1
grep -i "pair\|ideal\|lds_read" gpurun_out/r3_ubench.log
timeout 1200 python tools/kbench.py --fwd-only --dists encoder_like --out gpurun_out/r3_kbench1.json > gpurun_out/r3_kbench1.log 2>&1
grep -v "^bwd" gpurun_out/r3_kbench1.log | tail -60
timeout 400 tools/pmc_probe.sh r3_win_default fwd fwd_variant=12 > /dev/null 2>&1
timeout 400 tools/pmc_probe.sh r3_gather_hm fwd fwd_variant=3 fwd_head_major=1 > /dev/null 2>&1
cat gpurun_out/pmc_r3_win_default.txt gpurun_out/pmc_r3_gather_hm.txt
