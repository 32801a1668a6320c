"""Host-side helpers added in round 4 (no GPU): the container CPU quota, the padded-row zeroing node, the loss
normalisation without a read-back, the selector's options round trip."""
import io

import torch


def test_cpu_quota_reads_cgroup_v2_and_v1(monkeypatch):
    from memotr_amd.utils import host
    files = {}
    real_open = open

    def fake_open(path, *a, **k):
        if path in files:
            if files[path] is None:
                raise OSError(path)
            return io.StringIO(files[path])
        return real_open(path, *a, **k)

    monkeypatch.setattr("builtins.open", fake_open)
    monkeypatch.setattr(host.os, "sched_getaffinity", lambda _: set(range(256)), raising=False)
    files["/sys/fs/cgroup/cpu.max"] = "1600000 100000\n"
    assert host.cpu_quota() == 16.0                      # the MI355X boxes: 16 CPUs of quota on a 256-thread host
    files["/sys/fs/cgroup/cpu.max"] = "max 100000\n"
    assert host.cpu_quota() == 256.0
    files["/sys/fs/cgroup/cpu.max"] = None                # cgroup v1
    files["/sys/fs/cgroup/cpu/cpu.cfs_quota_us"] = "400000\n"
    files["/sys/fs/cgroup/cpu/cpu.cfs_period_us"] = "100000\n"
    assert host.cpu_quota() == 4.0
    files["/sys/fs/cgroup/cpu/cpu.cfs_quota_us"] = "-1\n"
    assert host.cpu_quota() == 256.0


def test_respect_cpu_quota_never_raises_the_thread_count(monkeypatch):
    from memotr_amd.utils import host
    before = torch.get_num_threads()
    try:
        monkeypatch.setattr(host, "cpu_quota", lambda: 4.0)
        torch.set_num_threads(2)
        assert host.respect_cpu_quota() == 2              # 4 * 0.5 = 2: nothing to do
        torch.set_num_threads(1)
        assert host.respect_cpu_quota() == 1              # never raised
        monkeypatch.setattr(host, "cpu_quota", lambda: 16.0)
        torch.set_num_threads(max(before, 2))
        assert host.respect_cpu_quota(processes=8) == 1   # eight ranks share the container
        monkeypatch.setenv("MEMOTR_NO_QUOTA_CAP", "1")
        torch.set_num_threads(2)
        assert host.respect_cpu_quota(processes=8) == 2
    finally:
        torch.set_num_threads(before)


def test_zero_rows_is_masked_fill_with_the_rows_known():
    """The padded rows of `value`, zeroed where the projection writes them: the reference's
    value.masked_fill(mask[..., None], 0) (models/ops/modules/ms_deform_attn.py:107-108), values and gradients."""
    from memotr_amd.modules.linear import long_linear
    from memotr_amd.modules.ms_deform_attn import _ZeroRows
    g = torch.Generator().manual_seed(0)
    for n in (40, 5000):                      # the row-linear path (a view of the product) and the split-K path (2-d, in place)
        x = torch.randn(2, n, 16, generator=g, requires_grad=True)
        w = torch.randn(16, 16, generator=g, requires_grad=True)
        b = torch.randn(16, generator=g, requires_grad=True)
        mask = torch.zeros(2, n, dtype=torch.bool)
        mask[0, 3] = mask[0, 7] = mask[1, n - 1] = True
        rows = mask.reshape(-1).nonzero().squeeze(1)
        got = long_linear(x, w, b, activation=lambda y: _ZeroRows.apply(y, rows))
        want = torch.nn.functional.linear(x, w, b).masked_fill(mask[..., None], 0.0)
        torch.testing.assert_close(got, want)
        go = torch.randn(2, n, 16, generator=g)
        ga = torch.autograd.grad(got, (x, w, b), go)
        gb = torch.autograd.grad(want, (x, w, b), go)
        for a_, b_ in zip(ga, gb):
            torch.testing.assert_close(a_, b_, rtol=1e-4, atol=3e-4)        # (fp32 sums over up to 10,000 rows, two orders)


def test_loss_normalisation_without_the_log_reads_nothing_back_and_is_the_same_loss():
    from memotr_amd.configs import dancetrack_config
    from memotr_amd.models.criterion import build as build_criterion
    crit = build_criterion(dancetrack_config(DEVICE="cpu", AVAILABLE_GPUS=""))
    crit.device = torch.device("cpu")
    crit.n_gts = [3, 0, 5]
    crit._acc = torch.tensor([[4.0, 1.0], [8.0, 2.0], [2.0, 3.0]])[:, : (2 if crit.aux_loss else 1)].contiguous()
    crit.log = {"frame0_box_l1_loss": torch.tensor(1.5), "frame1_box_l1_loss": torch.tensor(0.0),
                "frame2_box_l1_loss": torch.tensor(2.5)}
    with_log = crit.get_mean_by_n_gts()
    without = crit.get_mean_by_n_gts(with_log=False)
    assert without[1] == {}
    assert {k: float(v) for k, v in with_log[0].items()} == {k: float(v) for k, v in without[0].items()}
    assert float(with_log[0]["box_l1_loss"]) == 0.5 and float(with_log[0]["label_focal_loss"]) == 0.25        # / 8 boxes
    assert with_log[1]["frame0_box_l1_loss"] == (0.5, 1) and with_log[1]["frame1_box_l1_loss"] == (0.0, 1)
    assert with_log[1]["frame2_box_l1_loss"] == (0.5, 1)
