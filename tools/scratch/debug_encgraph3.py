import os, sys
os.environ["MEMOTR_REQUIRE_GRAPHS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_model_gpu import build_memotr_cuda
from model_helpers import small_config
from memotr_amd.engine import clip_forward_backward, make_synthetic_clip, clip_to_device
from memotr_amd.models.criterion import build as build_criterion
import memotr_amd.modules.ms_deform_attn as mod
MEM = []
def run(graphs, bf16, steps=1):
    if os.environ.get("FIX") == "gc":
        import gc as _gc
        _gc.collect(); torch.cuda.empty_cache()
    if os.environ.get("FIX") == "ws" and torch.cuda.graph.default_capture_stream is None:
        torch.cuda.graph.default_capture_stream = torch.cuda.Stream()
        with torch.cuda.stream(torch.cuda.graph.default_capture_stream):
            a = torch.randn(512, 512, device="cuda"); b = torch.randn(512, device="cuda")
            torch.addmm(b, a, a); torch.bmm(a[None], a[None]); (a @ a)
        torch.cuda.synchronize()
    os.environ["MEMOTR_ENCODE_GRAPHS"] = "1" if graphs else "0"
    torch.manual_seed(2)
    model = build_memotr_cuda(None, hidden=256, ffn=256, NUM_ENC_LAYERS=2, NUM_DEC_LAYERS=2).train()
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, mod.MSDeformAttn):
                m.sampling_offsets.weight.normal_(0, 0.02); m.attention_weights.weight.normal_(0, 0.05)
    cfg = small_config()
    cfg.update(HIDDEN_DIM=256, FFN_DIM=256, NUM_ENC_LAYERS=2, NUM_DEC_LAYERS=2, MATCH_COST_CLASS=2, MATCH_COST_BBOX=5,
               MATCH_COST_GIOU=2, LOSS_WEIGHT_FOCAL=2, LOSS_WEIGHT_L1=5, LOSS_WEIGHT_GIOU=2, AUX_LOSS_WEIGHT=[1.0], SAMPLE_LENGTHS=[2, 3, 4])
    criterion = build_criterion(cfg)
    batch = clip_to_device(make_synthetic_clip(clip_len=3, height=192, width=256, n_gts=5, seed=3), torch.device("cuda"))
    MEM.clear()
    orig = model.encode_frame
    import gc
    def snap():
        torch.cuda.synchronize()
        out = {}
        for o in gc.get_objects():
            try:
                if torch.is_tensor(o) and o.is_cuda and o.numel() > 0 and not isinstance(o, torch.nn.Parameter):
                    out[id(o)] = (o, tuple(o.shape), o.dtype, float(o.detach().double().sum()), o.data_ptr())
            except Exception:
                pass
        return out
    state = {}
    def rec(frame):
        mode = os.environ.get("PIN")
        if graphs and mode == "sync":
            torch.cuda.synchronize()
        if graphs and mode == "refs" and "r" in state and "done" not in state:
            state["done"] = 1
            gc.collect()
            lst = state["r"]
            seen = {}
            for i in range(len(lst)):
                rc = sys.getrefcount(lst[i])
                if rc <= 2:
                    o = lst[i]
                    k = (tuple(o.shape), str(o.dtype), o.requires_grad, o.grad_fn is not None, o._base is not None)
                    seen.setdefault(k, []).append(hex(o.data_ptr()))
                    del o
            for k, v in seen.items():
                print("   ORPHAN", k, len(v), v[:3])
        if graphs and os.environ.get("SNAP"):
            if "s" in state:
                now = snap()
                for k, (o, shp, dt, cs, ptr) in state["s"].items():
                    if k in now and now[k][0] is o and now[k][3] != cs:
                        print("   CHANGED", shp, dt, cs, "->", now[k][3], hex(ptr), "grad_fn", o.grad_fn is not None, "req", o.requires_grad)
        enc = orig(frame)
        if graphs and mode == "sync":
            torch.cuda.synchronize()
        if graphs and mode == "refs" and "r" not in state:
            state["r"] = [o for o in gc.get_objects() if torch.is_tensor(o) and o.is_cuda]
            print("   pinned", len(state["r"]))
        if graphs and os.environ.get("SNAP") and "s" not in state:
            state["s"] = snap()
        MEM.append({k: v.detach().clone() for k, v in enc.items() if torch.is_tensor(v)})
        return enc
    model.encode_frame = rec
    if os.environ.get("NO_DEC_GRAPHS"):
        from memotr_amd.models.decoder_graphs import DecoderGraphs
        DecoderGraphs.usable = lambda self, *a, **k: False
    for _ in range(steps):
        model.zero_grad()
        if bf16:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss, _ = clip_forward_backward(model, criterion, batch, torch.device("cuda"))
        else:
            loss, _ = clip_forward_backward(model, criterion, batch, torch.device("cuda"))
    return float(loss), {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}
from memotr_amd import _lib
for o in sys.argv[1:]:
    k, v = o.split("="); _lib.set_option(k, int(v)); print("option", k, v)
for bf16 in (False,):
    le, ge = run(False, bf16)
    me = [dict(m) for m in MEM]
    le2, ge2 = run(False, bf16)
    lg, gg = run(True, bf16, steps=2)
    mg = [dict(m) for m in MEM]
    for i, m in enumerate(mg):
        print("  encode call", i, {k: float((m[k].float() - me[0][k].float()).abs().max()) for k in m})
    lg1, gg1 = run(True, bf16, steps=1)
    lg3, gg3 = lg, gg
    print("bf16", bf16, "loss eager", le, "eager2", le2, "graph", lg, lg1, lg3)
    for tag, a in (("eager2", ge2), ("graph", gg), ("graph1", gg1), ("graph3", gg3)):
        rows = sorted(((float((a[n] - ge[n]).norm()) / (float(ge[n].norm()) + 1e-4), n) for n in ge), reverse=True)[:4]
        print("  ", tag, [(round(d, 4), n) for d, n in rows])
