#!/usr/bin/env python
"""Compile-time resource usage of every shipped kernel (hipcc -Rpass-analysis=kernel-resource-usage, gfx950): VGPRs,
AGPRs, SGPRs, spills, scratch, occupancy per SIMD, static LDS.  Runs without a GPU.

    python tools/kernel_resources.py > profiles/r03_kernel_resources.txt
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = ("VGPRs", "AGPRs", "TotalSGPRs", "VGPRs Spill", "SGPRs Spill", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]",
          "LDS Size [bytes/block]")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return out.stdout.splitlines()


def main():
    print("# hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage (compile-time facts of the shipped kernels;")
    print("# dynamic LDS -- the tiled / windowed / attention kernels -- is set at launch and not listed)")
    print(f"{'kernel':110s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'spill v/s':>10s} {'scratch':>8s} {'occ/SIMD':>8s} {'static LDS':>10s}")
    for src in ("msda_hip.hip", "clip_ops.hip"):
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics",
               "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(ROOT, "memotr_amd", "csrc", src), "-o", os.devnull]
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
        kernels, cur = [], None
        for line in err.splitlines():
            m = re.search(r"remark:\s+Function Name: (\S+)", line)
            if m:
                cur = {"name": m.group(1)}
                kernels.append(cur)
                continue
            m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
            if m and cur is not None and m.group(1).strip() in FIELDS:
                cur[m.group(1).strip()] = int(m.group(2))
        names = demangle([k["name"] for k in kernels])
        for k, n in zip(kernels, names):
            n = re.sub(r"\(anonymous namespace\)::", "", n)
            n = re.sub(r"\(.*$", "", n)[:110]
            print(f"{n:110s} {k.get('VGPRs', 0):5d} {k.get('AGPRs', 0):5d} {k.get('TotalSGPRs', 0):5d} "
                  f"{k.get('VGPRs Spill', 0):>6d}/{k.get('SGPRs Spill', 0):<3d} {k.get('ScratchSize [bytes/lane]', 0):8d} "
                  f"{k.get('Occupancy [waves/SIMD]', 0):8d} {k.get('LDS Size [bytes/block]', 0):10d}")


if __name__ == "__main__":
    sys.exit(main())
