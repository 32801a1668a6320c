#!/bin/bash
# Round 6 (late): the decoder capture's stacked (offsets; logits) projection -- a piece of the flat parameter's ONE split
# (MEMOTR_QPROJ_VIEW=1, default) against a concatenation inside the graph (=0); separate processes, one box.
#   bash tools/qproj_ab.sh   (first column: setting; ms per step; frames/s; host CPU ms per step)
export MEMOTR_BENCH_NO_KERNEL_LEGS=1
run() { name=$1; shift; env "$@" python bench.py --workload train --steps 10 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$name', round(d['ms_per_step'],2), round(d['value'],2), round(d['host_ms_per_step'],1))"; }
run piece MEMOTR_QPROJ_VIEW=1
run cat   MEMOTR_QPROJ_VIEW=0
run piece MEMOTR_QPROJ_VIEW=1
run cat   MEMOTR_QPROJ_VIEW=0
python tools/graph_census.py --encode-graphs 0 2>&1 | tail -5
