"""Bisect tool for the periodic 65-85 ms stall of the online-tracking loop (profiles/r03_infer_timeline.txt): a synthetic
frame made of the ingredients of the real one, each switchable, with the wall time of every iteration printed.  Not run
in round 3 (written after the GPU budget was spent); the real loop's facts so far: the stall sits in a blocking wait
right behind a device->host copy, the GPU is idle meanwhile, it needs frames issued back to back, and a bare
[matmuls, nonzero] loop does not show it.

    python tools/stall_repro.py                       # everything on
    python tools/stall_repro.py --no-small --no-memset   # leave ingredients out one by one
    python tools/stall_repro.py --read pinned         # how the host reads: nonzero | item | pageable | pinned

Ingredients per iteration: `--burst` ms of GEMMs (the encode half), `--small` tiny element-wise kernels (the query
updater / heads), `--memset` zero-fills, a device-to-device copy, a boolean-mask read (the tracker's births), a second
read after more small kernels (the result copy).
"""
import argparse
import time

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=45)
    ap.add_argument("--burst", type=float, default=8.0, help="ms of GEMM work per iteration")
    ap.add_argument("--small", type=int, default=120, help="tiny kernels per iteration")
    ap.add_argument("--memset", type=int, default=7)
    ap.add_argument("--read", default="nonzero", choices=["nonzero", "item", "pageable", "pinned", "none"])
    ap.add_argument("--no-small", action="store_true")
    ap.add_argument("--no-memset", action="store_true")
    ap.add_argument("--no-d2d", action="store_true")
    ap.add_argument("--second-read", action="store_true", default=True)
    ap.add_argument("--pause", type=float, default=0.0, help="ms of host sleep between iterations")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    a = torch.randn(2048, 2048, device=dev)
    small = [torch.randn(320, 256, device=dev) for _ in range(8)]
    big_src, big_dst = torch.randn(3, 800, 1344, device=dev), torch.empty(3, 800, 1344, device=dev)
    flag = torch.rand(300, device=dev) > 0.9
    fields = torch.randn(20, 8, device=dev, dtype=torch.float64)
    # calibrate the GEMM count for the requested burst
    for _ in range(5):
        (a @ a).sum().item()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        a @ a
    e.record()
    e.synchronize()
    per = s.elapsed_time(e) / 20
    n_gemm = max(1, int(round(args.burst / per)))

    def read(t):
        if args.read == "nonzero":
            return t.nonzero()
        if args.read == "item":
            return t.sum().item()
        if args.read == "pageable":
            return t.to("cpu")
        if args.read == "pinned":
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            ev.synchronize()
            return h
        return None

    ts = []
    for _ in range(args.iters):
        t0 = time.perf_counter()
        if not args.no_d2d:
            big_dst.copy_(big_src)
        for _ in range(n_gemm):
            a @ a
        if not args.no_small:
            x = small[0]
            for k in range(args.small // 2):
                x = x + small[k % 8]
        if not args.no_memset:
            for _ in range(args.memset):
                torch.zeros(4096, device=dev)
        read(flag)
        if not args.no_small:
            for k in range(args.small // 2):
                x = x * 0.5 + small[k % 8]
        if args.second_read and args.read != "none":
            read(fields if args.read in ("pageable", "pinned") else flag)
        ts.append((time.perf_counter() - t0) * 1e3)
        if args.pause:
            time.sleep(args.pause / 1e3)
    torch.cuda.synchronize()
    print(f"gemm {n_gemm} x {per:.2f} ms, small {0 if args.no_small else args.small}, memset "
          f"{0 if args.no_memset else args.memset}, read {args.read}, pause {args.pause} ms:")
    print(" ".join(f"{t:.1f}" for t in ts))
    body = sorted(ts[5:])
    print(f"median {body[len(body) // 2]:.2f} ms, max {body[-1]:.2f} ms, iterations over 3x the median: "
          f"{sum(t > 3 * body[len(body) // 2] for t in ts[5:])}")


if __name__ == "__main__":
    main()
