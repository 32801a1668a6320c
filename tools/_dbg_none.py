import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from model_helpers import load_model_golden, small_config, t
from test_model_gpu import build_memotr_cuda, _m6_setup
from memotr_amd.engine import clip_forward_backward
g = load_model_golden("M6_train_step")
model = build_memotr_cuda(g).train()
model.encode_chunks = "all"
criterion, batch = _m6_setup(g)
loss, _ = clip_forward_backward(model, criterion, batch, torch.device("cuda"))
print("captures", model.transformer.decoder.graphs().captures, "failed", model.transformer.decoder.graphs().failed)
for n, p in model.named_parameters():
    if p.requires_grad and p.grad is None:
        print("NONE", n)
