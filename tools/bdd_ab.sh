#!/bin/bash
# BDD100K fp32 step (config 5 at the reference's precision) under the package's switches, separate processes on one box
export MEMOTR_BENCH_NO_KERNEL_LEGS=1
run() { name=$1; st=$2; shift; shift; env "$@" python bench.py --config bdd100k --dtype f32 --steps $st --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$name', round(d['ms_per_step'],2), round(d['value'],2), round(d['host_ms_per_step'],1), d['decoder_graph_stats'])"; }
run steps6 6 A=1
run steps12 12 A=1
run steps24 24 A=1
