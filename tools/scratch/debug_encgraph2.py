import os, sys
os.environ["MEMOTR_REQUIRE_GRAPHS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from memotr_amd.configs import dancetrack_config
from memotr_amd.models import build_model
from memotr_amd.utils.nested_tensor import tensor_list_to_nested_tensor
from memotr_amd.utils.utils import set_seed
cfg = dancetrack_config(DEVICE="cuda", AVAILABLE_GPUS="0")
set_seed(42)
model = build_model(cfg).train()
g = torch.Generator().manual_seed(1)
nb = int(os.environ.get("NB", "5"))
frame = tensor_list_to_nested_tensor([torch.randn(3, 800, 1333, generator=g) for _ in range(nb)]).to("cuda")
w = torch.randn(1, 1, 256, device="cuda")
def one(graphs):
    os.environ["MEMOTR_ENCODE_GRAPHS"] = graphs
    model.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        enc = model(frame=frame, stage="encode")
    (enc["memory"].float() * w).sum().backward()
    torch.cuda.synchronize()
    return {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}
ref = one("0")
ref2 = one("0")
def cmp(a, b, tag):
    worst = []
    for n in a:
        d = float((a[n] - b[n]).norm()) / (float(b[n].norm()) + 1e-6)
        fin = bool(torch.isfinite(a[n]).all())
        if d > 0.05 or not fin:
            worst.append((n, round(d, 4), fin, float(a[n].abs().max())))
    print(tag, "params off by >5% or non-finite:", len(worst))
    for x in worst[:12]:
        print("    ", x)
cmp(ref2, ref, "eager vs eager")
for it in range(4):
    cmp(one("1"), ref, f"graph replay {it} vs eager")
