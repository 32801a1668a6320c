#!/usr/bin/env python
"""CPU model of the counting-sort backward's gather phase (msda_bwd_bins.h, phase 4): a wavefront walks the entry lists
of 16 cells together, its trip count is the longest of them.  For a sample of (region, head) pairs of the encoder-like
input: wave-iterations per pair and level with the cells in window order, sorted by list length, in a few length
classes, and the entries' share (no padding at all).  Round 5 used it to price a sorted cell list before building it.

    python tools/gather_sim.py
"""
import numpy as np, torch, sys
sys.path.insert(0, '/root/repo')
from memotr_amd.synth import make_inputs
x = make_inputs(device="cpu")
loc = x["loc"][0].numpy()        # (Lq, M, L, P, 2)
shapes = x["shapes"].numpy(); ls = x["level_start"].numpy()
L = 4; M = 8; P = 4
H = shapes[:,0]; W = shapes[:,1]
RY = (H[0] + 7)//8; RX = (W[0] + 7)//8
rng = np.random.default_rng(0)
def region_rows(ry, rx):
    rows = []
    for l in range(L):
        sh = L-1-l; side = 1 << sh
        for dy in range(side):
            for dx in range(side):
                py, px = (ry << sh) + dy, (rx << sh) + dx
                if py < H[l] and px < W[l]:
                    rows.append(ls[l] + py*W[l] + px)
    return rows
tot = {k: np.zeros(L) for k in ("unsorted", "sorted", "ideal", "cells", "entries", "classes6")}
nsamp = 0
for _ in range(150):
    ry, rx, m = rng.integers(RY), rng.integers(RX), rng.integers(M)
    rows = region_rows(ry, rx)
    for l in range(L):
        pts = loc[rows, m, l]          # (rows, P, 2)
        xs = pts[..., 0] * W[l] - 0.5; ys = pts[..., 1] * H[l] - 0.5
        w0 = np.floor(xs).astype(int); h0 = np.floor(ys).astype(int)
        cells = {}
        for dy in (0, 1):
            for dx in (0, 1):
                hh = h0 + dy; ww = w0 + dx
                ok = (hh >= 0) & (hh < H[l]) & (ww >= 0) & (ww < W[l])
                for a, b in zip(hh[ok].ravel(), ww[ok].ravel()):
                    cells[(a, b)] = cells.get((a, b), 0) + 1
        keys = sorted(cells)
        n = np.array([cells[k] for k in keys])
        def iters(order):
            # rounds of 64 cells, 4 waves x 16; per wave trips = ceil(max/2); LDS time ~ sum over waves of trips
            t = 0
            for r0 in range(0, len(order), 16):
                t += (order[r0:r0+16].max() + 1)//2
            return t
        tot["unsorted"][l] += iters(n)
        tot["sorted"][l] += iters(np.sort(n)[::-1])
        b = sum((n >= t).astype(int) for t in (32, 16, 8, 4, 2))       # six power-of-two length classes
        o = np.argsort(-b, kind="stable")
        tot["classes6"][l] += iters(n[o])
        tot["ideal"][l] += n.sum() / 32
        tot["cells"][l] += len(n); tot["entries"][l] += n.sum()
    nsamp += 1
for k, v in tot.items():
    print(f"{k:12s}", np.round(v / nsamp, 1), "sum", round(v.sum()/nsamp, 1))
