#!/usr/bin/env python
"""Pick the fastest rocBLAS / hipBLASLt solution for the pyramid-sized fp32 GEMMs of the encoder (run on the GPU box).

The library heuristics choose 256x256 macro-tiles for the (22323 x 256) x (256 x 256) projections, which
leaves two thirds of the 256 CUs idle (88 workgroups).  PyTorch's TunableOp benchmarks every solution of both
libraries for a GEMM the first time it sees it; this script runs one encoder layer (forward + backward) of the
given geometry with tuning on and writes the selections as a TunableOp CSV that memotr_amd.train_bench loads
read-only.  Usage: python tools/tune_gemm.py OUT.csv [H W] [max_ms_per_solution]
"""
import os
import sys
import time

import torch
import torch.cuda.tunable as tn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from memotr_amd.models.deformable_encoder import DeformableEncoderLayer  # noqa: E402
from memotr_amd.MultiScaleDeformableAttention import tag_host_shapes  # noqa: E402
from memotr_amd.synth import pyramid_shapes  # noqa: E402

out = sys.argv[1]
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (800, 1333)
max_ms = int(sys.argv[4]) if len(sys.argv) > 4 else 8

if os.path.exists(out):
    os.remove(out)
tn.enable(True)
tn.tuning_enable(True)
tn.set_filename(out, False)
tn.set_max_tuning_duration(max_ms)
tn.set_max_tuning_iterations(20)

dev = torch.device("cuda", 0)
shapes = pyramid_shapes(H, W)
S = sum(h * w for h, w in shapes)
spatial = torch.tensor(shapes, dtype=torch.long, device=dev)
tag_host_shapes(spatial, shapes)
lstart = torch.cat([spatial.new_zeros(1), (spatial[:, 0] * spatial[:, 1]).cumsum(0)[:-1]])
layer = DeformableEncoderLayer(d_model=256, d_ffn=2048, dropout=0.0, activation="ReLU", n_levels=4,
                                          n_heads=8, n_points=4).to(dev).train()
src = torch.randn(1, S, 256, device=dev, requires_grad=True)
pos = torch.randn(1, S, 256, device=dev)
ref = torch.rand(1, S, 4, 2, device=dev)
t0 = time.perf_counter()
for it in range(2):
    y = layer(src, pos, ref, spatial, lstart, None)
    y.square().mean().backward()
    torch.cuda.synchronize()
    print(f"pass {it}: {time.perf_counter() - t0:.1f} s, {len(tn.get_results())} tuned GEMMs", flush=True)

rows = tn.get_results()
with open(out + ".copy", "w") as f:           # own copy (the library also streams results to `out`)
    for k, v in tn.get_validators():
        f.write(f"Validator,{k},{v}\n")
    for r in rows:
        f.write(",".join(str(x) for x in r) + "\n")
for r in rows:
    print(r)
