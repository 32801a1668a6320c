import os, sys
os.environ["MEMOTR_REQUIRE_GRAPHS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from test_model_gpu import build_memotr_cuda
from memotr_amd.utils.nested_tensor import tensor_list_to_nested_tensor
import memotr_amd.modules.ms_deform_attn as mod
torch.manual_seed(2)
model = build_memotr_cuda(None, hidden=256, ffn=256, NUM_ENC_LAYERS=2, NUM_DEC_LAYERS=2).train()
frame = tensor_list_to_nested_tensor([torch.randn(3, 192, 256) for _ in range(3)]).to("cuda")
def stats(name, t):
    t = t.float()
    print(f"{name:30s} nan {int(torch.isnan(t).sum()):8d} inf {int(torch.isinf(t).sum()):6d} absmax {float(t[torch.isfinite(t)].abs().max()) if torch.isfinite(t).any() else -1:.4g}")
for graphs in ("0", "1"):
    os.environ["MEMOTR_ENCODE_GRAPHS"] = graphs
    for it in range(3):
        model.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            enc = model(frame=frame, stage="encode")
        mem = enc["memory"]
        stats(f"graphs={graphs} it={it} memory", mem)
        (mem.float() ** 2).mean().backward()
        bad = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        print("   dtype", mem.dtype, "params with non-finite grad:", len(bad), bad[:4])
        if it == 0 and graphs == "0":
            ref = mem.detach().float().clone()
        else:
            print("   max diff vs eager first:", float((mem.detach().float() - ref).abs().max()))
