#!/bin/bash
# SQ / LDS counters for one kernel variant:  tools/pmc_probe.sh fwd 6 2   -> gpurun_out/pmc_<op><variant>.txt
set -u
OP=$1; VAR=$2; MARGIN=${3:-2}
OUT=$PWD/gpurun_out/pmc_tmp; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES"
P2="SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $P -d "$OUT/p$i" -o p -- python tools/pmc_probe.py $OP $VAR $MARGIN > "$OUT/p$i.log" 2>&1
done
python - "$OUT" > gpurun_out/pmc_${OP}${VAR}_m${MARGIN}.txt <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "msda" in r["Kernel_Name"] and "jacobian" not in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("::")[-1].split("(")[0]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c in sorted(d):
        v = d[c]
        print(f"  {c:28s} {sum(v)/len(v):16.0f}")
PY
cat gpurun_out/pmc_${OP}${VAR}_m${MARGIN}.txt
rm -rf "$OUT"
