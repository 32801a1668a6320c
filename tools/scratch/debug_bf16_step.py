import os, sys
os.environ["MEMOTR_REQUIRE_GRAPHS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from memotr_amd.configs import dancetrack_config
from memotr_amd.engine import build_optimizer, clip_forward_backward, clip_to_device, make_synthetic_clip, optimizer_step
from memotr_amd.models import build_model
from memotr_amd.models.criterion import build as build_criterion
from memotr_amd.utils.utils import set_seed
cfg = dancetrack_config(DEVICE="cuda", AVAILABLE_GPUS="0")
dev = torch.device("cuda", 0)
for graphs in sys.argv[1:]:
    os.environ["MEMOTR_ENCODE_GRAPHS"] = graphs
    set_seed(42)
    model = build_model(cfg).train()
    criterion = build_criterion(cfg)
    opt = build_optimizer(cfg, model)
    batch = clip_to_device(make_synthetic_clip(5, 800, 1333, 10, seed=42), dev)
    for step in range(4):
        try:
            if os.environ.get("DBG_FP32"):
                loss, _ = clip_forward_backward(model, criterion, batch, dev)
            else:
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    loss, _ = clip_forward_backward(model, criterion, batch, dev)
        except Exception as e:
            print("graphs", graphs, "step", step, "EXC", type(e).__name__, e)
            break
        bad = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        gn = torch.sqrt(sum((p.grad.float() ** 2).sum() for p in model.parameters() if p.grad is not None))
        print("graphs", graphs, "step", step, "loss %.5f" % float(loss), "grad norm %.4f" % float(gn), "non-finite grads", len(bad), bad, flush=True)
        optimizer_step(model, opt, cfg["CLIP_MAX_NORM"])
    g = model.encode_graphs()
    print("   encode graphs: captures", g.captures, "replays", g.replays, "eager", g.eager)
    del model, opt
    torch.cuda.empty_cache()
