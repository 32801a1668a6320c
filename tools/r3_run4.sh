#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu -x > gpurun_out/r3_run4_tests.log 2>&1
tail -4 gpurun_out/r3_run4_tests.log
for j in 0.5 1.0 2.0 4.0; do JITTER=$j python tools/scratch/win_time.py "" "fwd_variant=3" 2>&1 | grep -v amdgpu.ids; done
python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
from bench import FusedCall, MsdaCall, time_kernel
from memotr_amd import _lib
from memotr_amd.synth import make_inputs
x = make_inputs(dist="encoder_like", device="cuda", batch=5)
call, fcall = MsdaCall(x), FusedCall(x)
for v in (0, 3):
    _lib.set_option("fwd_variant", v)
    r = []
    for c in (call, fcall):
        c.fwd(); torch.cuda.synchronize()
        r.append(time_kernel(c.fwd, iters=30) * 1e3)
    print("N=5 variant", v, "plain %.1f us fused %.1f us" % tuple(r), _lib.last_kernel())
PY
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3_bench_win.json 2> gpurun_out/r3_bench_win.err; cat gpurun_out/r3_bench_win.json
