#!/usr/bin/env python
"""Instruction histogram of one kernel in a hipcc -save-temps assembly file (CPU container; no GPU needed).

    python tools/isa_hist.py <file.s> <kernel-name-substring> [--dump out.s]
"""
import collections
import re
import sys


def kernel_body(lines, sub):
    start = None
    for i, l in enumerate(lines):
        if start is None and l.startswith("_ZN") and sub in l and ":" in l.split(";")[0]:
            start = i
        if start is not None and "s_endpgm" in l:
            return lines[start:i + 1]
    return None


def main():
    path, sub = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    b = kernel_body(lines, sub)
    if b is None:
        raise SystemExit(f"kernel *{sub}* not found")
    c = collections.Counter()
    for l in b:
        m = re.match(r"^\s+([a-z_0-9]+)\s", l)
        if m:
            c[m.group(1)] += 1
    print(sub, "static instructions:", sum(c.values()))
    groups = collections.Counter()
    for k, v in c.items():
        g = ("valu" if k.startswith("v_") else "salu" if k.startswith("s_") else "lds" if k.startswith("ds_") else
             "vmem" if k.startswith(("buffer_", "global_", "flat_", "scratch_")) else "other")
        groups[g] += v
    print("  " + ", ".join(f"{k}:{v}" for k, v in groups.most_common()))
    print("  " + ", ".join(f"{k}:{v}" for k, v in c.most_common(70)))
    if "--dump" in sys.argv:
        open(sys.argv[sys.argv.index("--dump") + 1], "w").write("\n".join(b))


if __name__ == "__main__":
    main()
