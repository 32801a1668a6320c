#!/bin/bash
# Fabric traffic per CALL of the encoder-shape fused forward and of the whole fused backward (counting-sort kernel + its
# two side kernels), the numbers bench.py quotes as `roofline.traffic` / `roofline_backward.traffic`:
#   tools/traffic_probe.sh [tag]   ->  gpurun_out/profiles/traffic.json, traffic_bwd.json, <tag>_traffic_probe.txt
# Separate --pmc passes for FETCH_SIZE and WRITE_SIZE (never combined with sys/hip/hsa tracing); FETCH_SIZE doubled per the
# gfx950 note of MI355X_MICROARCH.md.  Both files are stamped with the kernel label bench.py prints and the hash of the
# kernel sources, so a number taken on other kernels is recognised as stale.
set -u
TAG=${1:-r06}
OUT=$PWD/gpurun_out/traffic_tmp; rm -rf "$OUT"; mkdir -p "$OUT" gpurun_out/profiles
export TMPDIR=/tmp
for DIR in fwd bwd; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --output-format csv --pmc $C -d "$OUT/${DIR}_$C" -o p -- python tools/pmc_probe.py $DIR > "$OUT/${DIR}_$C.log" 2>&1
  done
done
python - "$OUT" "$TAG" <<'PY'
import collections, csv, glob, json, os, re, sys
root, tag = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.getcwd())
from memotr_amd.build import source_hash
lines = [f"# tools/traffic_probe.sh {tag}: fabric bytes per call (mean over the probe's launches; FETCH_SIZE x 2 on gfx950)"]
for d, out_name, key in (("fwd", "traffic.json", "msda_fwd_encoder_bytes_per_launch"),
                         ("bwd", "traffic_bwd.json", "msda_bwd_encoder_bytes_per_launch")):
    per = collections.defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = collections.defaultdict(list)
        for f in glob.glob(f"{root}/{d}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                # (the backward probe runs the forward once first -- the backward gets its output: not part of the call)
                if r["Counter_Name"] == c and "msda" in r["Kernel_Name"] and not (d == "bwd" and "msda_fwd" in r["Kernel_Name"]):
                    m = re.search(r"msda_\w+(<[^>]*>)?", r["Kernel_Name"])
                    vals[m.group(0) if m else r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
        for k, v in vals.items():
            per[k][c] = sum(v) / len(v)
    label = None
    for ln in open(f"{root}/{d}_FETCH_SIZE.log"):
        if ln.startswith("msda_"):
            label = ln.strip()
    fe = sum(v.get("FETCH_SIZE", 0.0) for v in per.values())
    wr = sum(v.get("WRITE_SIZE", 0.0) for v in per.values())
    total = int((2 * fe + wr) * 1024)
    lines.append(f"{d}: label {label}")
    for k, v in sorted(per.items()):
        lines.append(f"  {k:60s} FETCH {v.get('FETCH_SIZE', 0):10.0f} KiB (x2 = {2*v.get('FETCH_SIZE', 0)*1024/1e6:7.1f} MB)  "
                     f"WRITE {v.get('WRITE_SIZE', 0):10.0f} KiB ({v.get('WRITE_SIZE', 0)*1024/1e6:7.1f} MB)")
    lines.append(f"  whole call: 2 x FETCH {2*fe*1024/1e6:.1f} MB + WRITE {wr*1024/1e6:.1f} MB = {total/1e6:.1f} MB")
    json.dump({key: total, "kernel_label": label, "source_sha16": source_hash(), "fetch_size_KiB": fe, "write_size_KiB": wr,
               "kernels": {k: v for k, v in per.items()},
               "note": f"tools/traffic_probe.sh {tag}: every msda kernel of the call summed (round 6: the fused backward given the "
                       "forward's output is ONE counting-sort kernel + the zeroing launch); separate --pmc passes, FETCH_SIZE doubled (gfx950)"},
              open(os.path.join("gpurun_out", "profiles", out_name), "w"), indent=1)
open(os.path.join("gpurun_out", "profiles", f"{tag}_traffic_probe.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf "$OUT"
