#!/bin/bash
# Counters of one kernel configuration (encoder shape):   tools/pmc_probe.sh <tag> fwd|bwd [plain] [uniform] key=value ...
#   -> gpurun_out/pmc_<tag>.txt     (SQ / LDS passes + FETCH_SIZE / WRITE_SIZE / L2 hit passes, each its own run:
#   counters are never combined with sys/hip/hsa tracing)
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/pmc_tmp_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES"
P2="SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE"
P4="FETCH_SIZE"
P5="WRITE_SIZE"
P6="TCC_HIT_sum TCC_MISS_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $P -d "$OUT/p$i" -o p -- python tools/pmc_probe.py "$@" > "$OUT/p$i.log" 2>&1
done
python - "$OUT" "$*" > gpurun_out/pmc_${TAG}.txt <<'PY'
import csv, glob, re, sys, collections
root = sys.argv[1]
print("# tools/pmc_probe.sh", sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "msda" in r["Kernel_Name"]:
            m_ = re.search(r"msda_\w+(<[^>]*>)?", r["Kernel_Name"])      # (argument types hold "::" too)
            k = m_.group(0) if m_ else r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c in sorted(d):
        v = d[c]
        print(f"  {c:28s} {sum(v)/len(v):16.0f}")
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        f = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]); w = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
        # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-byte requests as 64 (MI355X_MICROARCH.md, HBM)
        print(f"  fabric bytes per launch: 2*FETCH {2*f*1024/1e6:.1f} MB + WRITE {w*1024/1e6:.1f} MB = {(2*f+w)*1024/1e6:.1f} MB")
    if "TCC_HIT_sum" in d:
        h = sum(d["TCC_HIT_sum"]); m = sum(d["TCC_MISS_sum"])
        print(f"  L2 hit rate {h/(h+m):.3f}")
PY
cat gpurun_out/pmc_${TAG}.txt
rm -rf "$OUT"
