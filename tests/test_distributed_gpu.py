"""GPU: the data-parallel train step on the RCCL (``nccl``) backend.

  * world size 1 (any GPU box): the real model wrapped in DistributedDataParallel on ``nccl``, two clip steps
    with the encode/decode split -- so the first multi-GPU run is not also the first NCCL run;
  * world size 2 (only when two devices are visible): replicas stay bit-identical after two steps.
Workers run in spawned processes (one per GPU, like ``torch.distributed.run`` launches bench.py).
"""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from memotr_amd import _lib
    from memotr_amd.configs import dancetrack_config
    from memotr_amd.engine import (build_optimizer, clip_forward_backward, clip_to_device, make_synthetic_clip,
                                   optimizer_step)
    from memotr_amd.models import build_model
    from memotr_amd.models.criterion import build as build_criterion
    cfg = dancetrack_config(DEVICE="cuda", AVAILABLE_GPUS=str(rank), NUM_ENC_LAYERS=2, NUM_DEC_LAYERS=2, FFN_DIM=512)
    torch.manual_seed(100 + rank)                    # DDP must broadcast rank 0's weights
    model = build_model(cfg).train()
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[rank], find_unused_parameters=False)
    criterion = build_criterion(cfg)
    opt = build_optimizer(cfg, ddp)
    batch = clip_to_device(make_synthetic_clip(clip_len=3, height=256, width=320, n_gts=3 + rank, seed=7 + rank), dev)
    losses = []
    for _ in range(2):                               # a second step raises if a bucket was left unreduced
        loss, _ = clip_forward_backward(ddp, criterion, batch, dev)
        assert all(p.grad is not None for p in ddp.parameters() if p.requires_grad)
        optimizer_step(ddp, opt, cfg["CLIP_MAX_NORM"])
        losses.append(float(loss))
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1) for p in ddp.parameters()]).cpu()
    torch.save({"flat": flat, "losses": losses, "kernel": _lib.last_kernel(),
                "backend": dist.get_backend(), "world": dist.get_world_size()}, f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def _run(world, tmp_path):
    port = _free_port()
    out = str(tmp_path / "rank")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    return [torch.load(f"{out}.{r}") for r in range(world)]


def test_single_rank_nccl_ddp_two_clip_steps(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    (r0,) = _run(1, tmp_path)
    assert r0["backend"] == "nccl" and r0["world"] == 1
    assert all(l == l and abs(l) < 1e6 for l in r0["losses"])
    assert torch.isfinite(r0["flat"]).all()
    assert "msda" in r0["kernel"]


def test_two_rank_nccl_replicas_stay_in_sync(tmp_path):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    r0, r1 = _run(2, tmp_path)
    assert r0["world"] == 2 and torch.equal(r0["flat"], r1["flat"]), "replicas diverged"
    assert r0["losses"][0] != r1["losses"][0]


# ----------------------------------------------------------------------------- two ranks, ONE device, gloo
def _worker_shared_device(rank, world, port, out_path):
    """Both ranks drive cuda:0 and exchange over gloo: every N > 1 code path of the train step (DDP bucket hooks on
    device tensors with gradient_as_bucket_view, the decoder graphs' flat-parameter cat under the hooks, the
    ground-truth-count all-reduce, T partial forwards per backward) meets real device tensors without a second GPU."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", MEMOTR_REQUIRE_GRAPHS="1")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from memotr_amd.configs import dancetrack_config
    from memotr_amd.engine import (build_optimizer, clip_forward_backward, clip_to_device, make_synthetic_clip,
                                   optimizer_step)
    from memotr_amd.models import build_model
    from memotr_amd.models.criterion import build as build_criterion
    cfg = dancetrack_config(DEVICE="cuda", AVAILABLE_GPUS="0", NUM_ENC_LAYERS=1, NUM_DEC_LAYERS=2, FFN_DIM=256)
    torch.manual_seed(100 + rank)                    # DDP must broadcast rank 0's weights
    from memotr_amd.models.memotr import build as build_memotr      # build_model() would pick cuda:<rank>
    from memotr_amd.modules.linear import configure_blas
    configure_blas()
    model = build_memotr(config=cfg).to(dev).train()
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], find_unused_parameters=False,
                                                    gradient_as_bucket_view=True, broadcast_buffers=False)
    criterion = build_criterion(cfg)
    opt = build_optimizer(cfg, ddp)
    # the ranks differ in ground-truth count (query buckets) AND in clip length (T partial forwards per backward, the
    # length of the count vector): every collective of the step -- the count all-reduce, DDP's gradient buckets --
    # must still be met by both
    clip_len = 3 - rank
    batch = clip_to_device(make_synthetic_clip(clip_len=clip_len, height=192, width=256, n_gts=3 + 2 * rank, seed=7 + rank), dev)
    losses, counts, step_ms = [], None, []
    import time
    for _ in range(2):                               # a second step raises if a bucket was left unreduced
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss, _ = clip_forward_backward(ddp, criterion, batch, dev)
        assert all(p.grad is not None for p in ddp.parameters() if p.requires_grad)
        counts = list(criterion.n_gts)
        optimizer_step(ddp, opt, cfg["CLIP_MAX_NORM"])
        losses.append(float(loss))
        step_ms.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    g = model.transformer.decoder.graphs()
    flat = torch.cat([p.detach().reshape(-1) for p in ddp.parameters()]).cpu()
    from memotr_amd.utils.host import cpu_quota, respect_cpu_quota
    threads = respect_cpu_quota(processes=world)       # what bench.py's self-launched ranks do (LOCAL_WORLD_SIZE)
    torch.save({"flat": flat, "losses": losses, "captures": g.captures, "replays": g.replays, "eager": g.eager,
                "n_gts": counts, "backend": dist.get_backend(), "world": dist.get_world_size(), "clip_len": clip_len,
                "step_ms": step_ms, "threads": threads, "quota": cpu_quota()}, f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_device_over_gloo_stay_in_sync(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    port = _free_port()
    out = str(tmp_path / "shared")
    mp.spawn(_worker_shared_device, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = (torch.load(f"{out}.{r}") for r in range(2))
    assert r0["world"] == 2 and r0["backend"] == "gloo"
    assert torch.equal(r0["flat"], r1["flat"]), "replicas diverged"
    assert torch.isfinite(r0["flat"]).all()
    assert r0["losses"][0] != r1["losses"][0]              # the ranks saw different clips
    # ranks with different ground-truth counts (3 and 5 tracks -> different query buckets) still meet at every
    # all-reduce: the second (steady-state) step of the two ranks ends within a few milliseconds of each other.  The
    # spread is printed for the record (both ranks share ONE device here, so it is an upper bound on rank skew)
    # -- and different clip lengths (3 and 2 frames): the step ends at DDP's last bucket on both, so the shorter clip's
    # rank waits for the longer one; rank_skew_ms is what bench.py reports for a real multi-GPU run
    spread = abs(r0["step_ms"][1] - r1["step_ms"][1])
    print(f"per-rank step time (ms): rank0 {r0['step_ms'][1]:.1f} rank1 {r1['step_ms'][1]:.1f} rank_skew_ms {spread:.1f}")
    assert spread < 0.5 * max(r0["step_ms"][1], r1["step_ms"][1])
    assert (r0["clip_len"], r1["clip_len"]) == (3, 2) and len(r0["n_gts"]) == 3 and len(r1["n_gts"]) == 2
    for r in (r0, r1):                                     # decoder graphs active under DDP: T frames x 2 steps
        assert r["captures"] == r["clip_len"] and r["replays"] == 2 * r["clip_len"] and r["eager"] == 0, r
        # the thread pools are sized to this rank's share of the container's CPU quota (utils/host.py)
        assert 1 <= r["threads"] <= max(1, int(r["quota"] * 0.5 / 2) + 1), (r["threads"], r["quota"])
