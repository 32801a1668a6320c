"""`python bench.py --gpus N` must be runnable as is: with no launcher around it (WORLD_SIZE unset) it starts one rank
per GPU itself -- the reference's launch line is `python -m torch.distributed.run --nproc_per_node=8 main.py`
(README.md:104, main.py:100-101).  No GPU here: the device count is stubbed through MEMOTR_BENCH_DEVICE_COUNT and the
ranks stop at the "needs a GPU" check, which is far enough to prove the spawn path parses and forwards the arguments."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    e.update(env)
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=300)


def test_gpus_2_without_devices_fails_on_the_device_count_not_on_a_launcher():
    r = _run(["--gpus", "2"], MEMOTR_BENCH_DEVICE_COUNT="1")
    assert r.returncode != 0
    assert "only 1 GPU(s) visible" in r.stderr and "device count" in r.stderr, r.stderr[-400:]
    assert "launch with" not in r.stderr


def test_dry_launch_prints_the_rank_launcher_command_with_the_script_arguments():
    r = _run(["--gpus", "4", "--steps", "3", "--warmup", "1"], MEMOTR_BENCH_DEVICE_COUNT="8", MEMOTR_BENCH_DRY_LAUNCH="1")
    assert r.returncode == 0, r.stderr[-400:]
    argv = json.loads(r.stdout.strip().splitlines()[-1])["launch"]
    assert argv[1:3] == ["-m", "torch.distributed.run"]
    assert argv[argv.index("--nproc-per-node") + 1] == "4"
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    tail = argv[argv.index(BENCH) + 1:]
    assert tail == ["--gpus", "4", "--steps", "3", "--warmup", "1"]


def test_self_launch_starts_one_rank_per_gpu_and_each_reaches_the_gpu_check():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], MEMOTR_BENCH_DEVICE_COUNT="2")
    assert r.returncode != 0                       # no GPU in this container: the ranks stop at the device check
    assert r.stderr.count("bench.py needs a GPU") == 2, r.stderr[-800:]


def test_a_launcher_that_started_the_wrong_world_is_reported_as_such():
    r = _run(["--gpus", "2"], WORLD_SIZE="3", RANK="0", LOCAL_RANK="0", MEMOTR_BENCH_DEVICE_COUNT="4")
    assert r.returncode != 0
    assert "WORLD_SIZE=3" in r.stderr or "needs a GPU" in r.stderr


def test_eight_rank_launch_rehearsal_rendezvous_thread_pools_and_one_json_line():
    """Round-5 verdict, item 7: the 8-GPU launch minus the GPUs.  `python bench.py --gpus 8` with a stubbed device count
    starts eight ranks itself; they rendezvous on 127.0.0.1, size their torch thread pools by an eighth of the
    container's CPU quota (eight ranks share it on the driver's box: 16 CPUs -> 2 per rank, one of them for the pool),
    all-reduce once, and exactly one JSON line comes out -- rank 0's."""
    r = _run(["--gpus", "8", "--steps", "1", "--warmup", "0"], MEMOTR_BENCH_DEVICE_COUNT="8", MEMOTR_BENCH_REHEARSAL="1")
    assert r.returncode == 0, r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-800:]
    out = json.loads(lines[0])
    assert out["rehearsal"] and out["n_gpus"] == 8 and out["local_world"] == 8
    assert out["rank_sum"] == 36.0                                    # 1 + 2 + ... + 8: every rank took part
    from memotr_amd.utils.host import cpu_quota
    want = max(1, int(cpu_quota() * 0.5 / 8))
    assert out["torch_threads"] == want and out["torch_threads_sum"] == 8.0 * want, out
    assert abs(out["cpu_quota_per_rank"] - cpu_quota() / 8) < 1e-9
