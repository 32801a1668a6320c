"""clipops_assign_f32 against scipy on the matcher's problem shape: six decoder layers x (310 queries, T ground truths).
HIP-event time of the kernel vs host time of the six scipy calls (+ the device->host copy they need)."""
import os
import sys
import time

import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from memotr_amd.functions import clip_ops  # noqa: E402

g = torch.Generator().manual_seed(0)
for T in (5, 10, 20):
    cost = torch.randn(6, 310, T, generator=g).cuda()
    for _ in range(5):
        clip_ops.assign(cost)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50):
        clip_ops.assign(cost)
    e.record(); e.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        c = cost.cpu().numpy()
        for layer in range(6):
            linear_sum_assignment(c[layer])
    host = (time.perf_counter() - t0) / 50 * 1e6
    print(f"6 x (310, {T}): device kernel {s.elapsed_time(e) / 50 * 1e3:.1f} us; host: copy + six scipy calls {host:.1f} us")
