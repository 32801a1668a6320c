import os, sys, torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, "/root/repo")
from memotr_amd.configs import dancetrack_config
from memotr_amd.engine import build_optimizer, clip_forward_backward, clip_to_device, make_synthetic_clip, optimizer_step
from memotr_amd.models import build_model
from memotr_amd.models.criterion import build as build_criterion
cfg = dancetrack_config(DEVICE="cuda", AVAILABLE_GPUS="0")
dev = torch.device("cuda", 0)
model = build_model(cfg).train(); criterion = build_criterion(cfg); opt = build_optimizer(cfg, model)
batch = clip_to_device(make_synthetic_clip(5, 800, 1333, 10, seed=42), dev)
def step():
    clip_forward_backward(model, criterion, batch, dev); optimizer_step(model, opt, 0.1)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
rows = [e for e in ka if e.key in ("aten::mm", "aten::addmm", "aten::bmm", "aten::convolution_backward", "aten::miopen_convolution", "aten::_scaled_dot_product_flash_attention", "aten::linear")]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:45]:
    print(f"{e.device_time_total/1e3:8.2f} ms n={e.count:4d} avg {e.device_time_total/e.count:8.1f} us cpu {e.cpu_time_total/e.count:7.1f} us  {e.key:28s} {str(e.input_shapes)[:110]}")
print("all ops by shape, self device time:")
allops = sorted(ka, key=lambda e: -e.self_device_time_total)
for e in allops[:70]:
    print(f"{e.self_device_time_total/1e3:8.2f} ms n={e.count:4d} avg {e.self_device_time_total/max(e.count,1):8.1f} us  {e.key[:42]:42s} {str(e.input_shapes)[:100]}")
print("copies by shape (device time):")
cp = [e for e in ka if e.key in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::cat", "aten::_to_copy", "aten::fill_", "aten::zero_", "aten::add_", "aten::add", "aten::mul", "aten::sum")]
cp.sort(key=lambda e: -e.self_device_time_total)
for e in cp[:40]:
    print(f"{e.self_device_time_total/1e3:8.2f} ms n={e.count:4d} avg {e.self_device_time_total/max(e.count,1):8.1f} us  {e.key:18s} {str(e.input_shapes)[:120]}")
# host-side view: top ops by self CPU time
ka2 = sorted(prof.key_averages(), key=lambda e: -e.self_cpu_time_total)
print("top self-CPU ops:")
for e in ka2[:30]:
    print(f"{e.self_cpu_time_total/1e3:8.2f} ms n={e.count:5d} avg {e.self_cpu_time_total/e.count:7.1f} us  {e.key[:70]}")
