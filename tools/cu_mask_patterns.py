#!/usr/bin/env python
"""Which CUs does a bit of hipExtStreamCreateWithCUMask's mask stand for on this part?  GEMM rate and the latency of a
small-kernel chain on another stream, for a list of mask patterns (tools/cu_mask_probe.py found 128..240 low bits all at
half rate)."""
import ctypes
import time

import torch

hip = ctypes.CDLL("libamdhip64.so")


def stream_of(words):
    arr = (ctypes.c_uint32 * len(words))(*words)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(len(words)), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


dev = torch.device("cuda", 0)
a = torch.randn(8192, 4096, device=dev)
b = torch.randn(4096, 4096, device=dev)
out = torch.empty(8192, 4096, device=dev)
x = torch.zeros(320, 256, device=dev)
for _ in range(3):
    torch.mm(a, b, out=out)
torch.cuda.synchronize()


def rate(st, n=10):
    st.wait_stream(torch.cuda.current_stream())
    best = 1e9
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(st):
            for _ in range(n):
                torch.mm(a, b, out=out)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def chain_next_to(st, n=20):
    main = torch.cuda.current_stream()
    torch.cuda.synchronize()
    st.wait_stream(main)
    with torch.cuda.stream(st):
        for _ in range(n):
            torch.mm(a, b, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        x.add_(1.0)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


F = 0xFFFFFFFF
full = rate(torch.cuda.Stream())
print(f"unmasked stream: {full:.2f} ms per 10 GEMMs")
pats = {
    "8 words, all ones": [F] * 8,
    "8 words, bit 255 clear": [F] * 7 + [0x7FFFFFFF],
    "8 words, bit 0 clear": [0xFFFFFFFE] + [F] * 7,
    "8 words, bits 248-255 clear (top byte)": [F] * 7 + [0x00FFFFFF],
    "8 words, bits 0-7 clear": [0xFFFFFF00] + [F] * 7,
    "8 words, word 7 = 0": [F] * 7 + [0],
    "8 words, word 0 = 0": [0] + [F] * 7,
    "8 words, top byte of every word clear": [0x00FFFFFF] * 8,
    "8 words, bit 31 of every word clear": [0x7FFFFFFF] * 8,
    "8 words, even words only": [F, 0] * 4,
    "8 words, low 16 bits of every word": [0x0000FFFF] * 8,
    "4 words, all ones": [F] * 4,
    "9 words, all ones": [F] * 9,
    "10 words, all ones": [F] * 10,
    "10 words, word 9 = 0": [F] * 9 + [0],
    "16 words, all ones": [F] * 16,
    "16 words, words 14-15 = 0": [F] * 14 + [0, 0],
    "1 word, all ones": [F],
    "2 words, all ones": [F, F],
}
for name, words in pats.items():
    try:
        st = stream_of(words)
        r = rate(st)
        c = chain_next_to(st)
        print(f"{name:45s}: {r:7.2f} ms ({full / r * 100:5.1f} %)   chain of 100 adds next to it: {c:7.2f} ms")
    except AssertionError as e:
        print(f"{name:45s}: create failed rc={e}")
