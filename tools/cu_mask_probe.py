#!/usr/bin/env python
"""Round 6: can a chain of small dependent kernels run next to large GEMMs without being stretched to their length?

Round 2 measured it on two ordinary streams (tools/stream_overlap_probe.py): a 2.6 ms chain took 16 ms next to 15 ms of
GEMMs, at either priority -- every link waits for a workgroup of the large kernel to retire.  Here the large kernels run
on a stream created with a CU mask (hipExtStreamCreateWithCUMask) that leaves some CUs out; the chain's stream may use
every CU, so the ones left out are always free for it.

  1. GEMM time on a masked stream vs the number of mask bits set (which bits map to which CUs is not documented here:
     throughput ~ bits set is the check that the mask is honoured at all);
  2. a chain of tiny dependent kernels (and a chain with decoder-sized GEMMs) alone / next to GEMMs on an unmasked side
     stream / next to GEMMs on masked side streams.
"""
import ctypes
import sys
import time

import torch

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits_set: int, total: int = 256, high: bool = False):
    """A stream whose kernels may run on ``bits_set`` of ``total`` CUs (the low bits of the mask, or the high ones)."""
    words = total // 32
    mask = [0] * words
    for i in range(bits_set):
        j = (total - 1 - i) if high else i
        mask[j // 32] |= 1 << (j % 32)
    arr = (ctypes.c_uint32 * words)(*mask)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(words), arr)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(s.value), s


def timed(fn, stream=None, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


dev = torch.device("cuda", 0)
props = torch.cuda.get_device_properties(0)
print("device:", props.name, "CUs:", props.multi_processor_count)
a = torch.randn(8192, 4096, device=dev)
b = torch.randn(4096, 4096, device=dev)
out = torch.empty(8192, 4096, device=dev)


def gemms(n):
    for _ in range(n):
        torch.mm(a, b, out=out)


gemms(3)
print("\n# 1. ten 8192x4096x4096 fp32 GEMMs on a masked stream")
base = None
for bits in (256, 240, 224, 192, 128, 64):
    st, _h = masked_stream(bits)
    st.wait_stream(torch.cuda.current_stream())
    def run():
        with torch.cuda.stream(st):
            gemms(10)
    ms = timed(run)
    base = base or ms
    print(f"  {bits:3d} bits: {ms:7.2f} ms   ({base / ms * 100:5.1f} % of the full rate; bits/256 = {bits / 256 * 100:.0f} %)")

print("\n# 2. a chain of dependent small kernels next to GEMMs")
x = torch.zeros(320, 256, device=dev)
w = torch.randn(256, 256, device=dev) * 0.01
y = torch.empty(320, 256, device=dev)
g = torch.cuda.CUDAGraph()
side0 = torch.cuda.Stream()
side0.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side0):
    for _ in range(3):
        for _ in range(10):
            x.add_(1.0)
            torch.mm(x, w, out=y)
torch.cuda.current_stream().wait_stream(side0)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    for _ in range(150):            # 300 kernels: tiny add, decoder-sized GEMM
        x.add_(1.0)
        torch.mm(x, w, out=y)


def chain():
    g.replay()


alone = timed(chain, reps=5)
print(f"  chain alone (graph of 300 kernels): {alone:.2f} ms")
n_gemm = 40
gemm_alone = timed(lambda: gemms(n_gemm))
print(f"  {n_gemm} GEMMs alone: {gemm_alone:.2f} ms")


def both(side):
    def run():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            gemms(n_gemm)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        chain()
        ev1.record()
        torch.cuda.synchronize()
        run.chain_ms = ev0.elapsed_time(ev1)
    return run


for label, side in (("unmasked side stream", torch.cuda.Stream()),
                    ("unmasked side, chain on a high-priority stream", None),
                    ("side masked to 240 CUs", masked_stream(240)[0]),
                    ("side masked to 224 CUs", masked_stream(224)[0]),
                    ("side masked to 192 CUs", masked_stream(192)[0]),
                    ("side masked to 224 CUs (high bits)", masked_stream(224, high=True)[0])):
    if side is None:
        hp = torch.cuda.Stream(priority=-1)
        sd = torch.cuda.Stream()
        def run():
            main = torch.cuda.current_stream()
            sd.wait_stream(main)
            hp.wait_stream(main)
            with torch.cuda.stream(sd):
                gemms(n_gemm)
            with torch.cuda.stream(hp):
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
                chain()
                ev1.record()
            torch.cuda.synchronize()
            run.chain_ms = ev0.elapsed_time(ev1)
        r = run
    else:
        r = both(side)
    total = timed(r, reps=3)
    print(f"  {label:50s}: chain {r.chain_ms:7.2f} ms, everything {total:7.2f} ms (GEMMs alone {gemm_alone:.2f})")
sys.stdout.flush()
