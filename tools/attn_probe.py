#!/usr/bin/env python
"""Decoder self-attention (B=1, L=320, 8 heads of 32): the hand-written kernels against
F.scaled_dot_product_attention (AOTriton), forward and forward+backward, HIP-event timing.  Run on the GPU box."""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from memotr_amd.functions import clip_ops  # noqa: E402

B, L, H, E = 1, 320, 8, 256
qk = torch.randn(B, L, 2 * E, device="cuda", requires_grad=True)
v = torch.randn(B, L, E, device="cuda", requires_grad=True)
mask = torch.zeros(B, L, dtype=torch.bool, device="cuda")
mask[:, 310:] = True
up = torch.randn(B, L, E, device="cuda")


def sdpa(qk_p, v_p, m, h):
    q, k = (t.transpose(1, 2) for t in qk_p.view(B, L, 2, h, E // h).unbind(2))
    vh = v_p.view(B, L, h, E // h).transpose(1, 2)
    out = F.scaled_dot_product_attention(q, k, vh, attn_mask=None if m is None else ~m.view(B, 1, 1, L))
    return out.transpose(1, 2).reshape(B, L, E)


def timed(fn, iters=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for name, fn in (("hand-written kernels", clip_ops.self_attention), ("scaled_dot_product_attention", sdpa)):
    for m in (None, mask):
        with torch.no_grad():
            f = timed(lambda: fn(qk, v, m, H))

        def fb():
            qk.grad = v.grad = None
            (fn(qk, v, m, H) * up).sum().backward()
        print(f"{name:30s} mask={'yes' if m is not None else 'no ':3s}  forward {f:7.1f} us   forward+backward (incl. autograd) {timed(fb, 100):7.1f} us")

# device time per kernel (torch.profiler), one forward + backward of each path
from torch.profiler import ProfilerActivity, profile  # noqa: E402
for name, fn in (("hand-written kernels", clip_ops.self_attention), ("scaled_dot_product_attention", sdpa)):
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(10):
            qk.grad = v.grad = None
            (fn(qk, v, mask, H) * up).sum().backward()
        torch.cuda.synchronize()
    print(name)
    for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:8]:
        print(f"   {e.device_time_total / max(e.count, 1):7.1f} us x{e.count // 10:2d}  {e.key[:90]}")
