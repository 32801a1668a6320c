#!/usr/bin/env python
"""The one-launch Linear backward (clipops_linear_bwd_f32) against the chain it replaces ([threshold_backward,] two
GEMMs, column sum), eagerly and replayed from a hipGraph (the decoder's situation), at the decoder's shapes.

    python tools/linear_bwd_probe.py [--out gpurun_out/linear_bwd_probe.txt]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from memotr_amd.functions import clip_ops  # noqa: E402
from memotr_amd.modules.linear import configure_blas  # noqa: E402


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


def graphed(fn, reps=20):
    """`reps` back-to-back calls captured in one graph: what a dependent chain costs inside a replay."""
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        for _ in range(reps):
            fn()
    return lambda: g.replay(), reps


def main():
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else "gpurun_out/linear_bwd_probe.txt"
    try:
        configure_blas()
    except Exception:  # noqa: BLE001
        pass
    lines = []
    for rows, in_f, out_f, relu in ((320, 256, 256, False), (320, 256, 256, True), (320, 512, 256, True),
                                    (320, 256, 1024, True), (320, 1024, 256, False), (320, 256, 4, False),
                                    (320, 256, 768, False), (10, 256, 256, False)):
        x = torch.randn(rows, in_f, device="cuda")
        w = torch.randn(out_f, in_f, device="cuda") / in_f ** 0.5
        gy = torch.randn(rows, out_f, device="cuda")
        y = torch.relu(x @ w.t()) if relu else None

        def chain():
            g2 = torch.ops.aten.threshold_backward(gy, y, 0.0) if relu else gy
            return g2 @ w, g2.t() @ x, clip_ops.colsum(g2)

        def one():
            return clip_ops.linear_bwd(gy, y, x, w)

        bias = torch.randn(out_f, device="cuda")

        def lib_fwd():
            return torch._addmm_activation(bias, x, w.t(), use_gelu=False) if relu else torch.addmm(bias, x, w.t())

        def one_fwd():
            return clip_ops.linear_fwd(x, w, bias, relu)

        gf, reps_f = graphed(lib_fwd)
        gof, _ = graphed(one_fwd)
        fwd_line = (f"   ||  forward: library {timed(lib_fwd):5.1f} us  tile kernel {timed(one_fwd):5.1f} us  | in a graph "
                    f"{timed(gf, 50) / reps_f:5.1f} vs {timed(gof, 50) / reps_f:5.1f} us") if in_f % 4 == 0 else ""
        t_chain, t_one = timed(chain), timed(one)
        gc, reps = graphed(chain)
        go, _ = graphed(one)
        tg_chain, tg_one = timed(gc, 50) / reps, timed(go, 50) / reps
        lines.append(f"rows {rows:4d} in {in_f:4d} out {out_f:4d} relu {int(relu)}:  eager chain {t_chain:6.1f} us  one launch "
                     f"{t_one:6.1f} us   |  inside a graph: chain {tg_chain:6.1f} us  one launch {tg_one:6.1f} us" + fwd_line)
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
