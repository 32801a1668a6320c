#!/usr/bin/env python
"""bf16-storage fused forward at the encoder shape: the windowed kernel on 64-byte rows (round 6) against the gather kernel
(rounds 2-5), both pyramids, N = 1 and N = 5:   python tools/bf16_fwd_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import FusedCall, FusedCallBf16, time_kernel  # noqa: E402
from memotr_amd import _lib  # noqa: E402
from memotr_amd.synth import make_inputs  # noqa: E402

print("pyramid batch | fp32 win us | bf16 gather us (frac of 8 TB/s on its bytes) | bf16 win us (frac)")
for name, (h, w) in (("800x1333", (800, 1333)), ("720x1280", (720, 1280))):
    for batch in (1, 5):
        x = make_inputs(height=h, width=w, device="cuda", batch=batch)
        f32, b16 = FusedCall(x), FusedCallBf16(x)
        _lib.set_call_site(100 + batch)
        t32 = time_kernel(f32.fwd) * 1e3
        row = []
        for on in (0, 1):
            _lib.set_option("fwd_win_bf16", on)
            _lib.set_call_site(200 + 10 * on + batch)
            t = time_kernel(b16.fwd) * 1e3
            row.append((t, b16.bytes() / (t * 1e-6) / 8e12, _lib.last_kernel()))
        _lib.set_option("fwd_win_bf16", 1)
        print(f"{name} N={batch} | {t32:7.1f} | {row[0][0]:7.1f} ({row[0][1]:.3f}) {row[0][2]} | {row[1][0]:7.1f} ({row[1][1]:.3f}) {row[1][2]}")
