#!/bin/bash
# does the warm-up length change a line?  (late graph captures would show as a faster step after a longer warm-up)
export MEMOTR_BENCH_NO_KERNEL_LEGS=1
run() { name=$1; shift; python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); g=d['decoder_graph_stats']; print('$name', round(d['ms_per_step'],2), round(d['value'],2), 'decoder captures', g['captures'], 'updater captures', g.get('updater_captures'))"; }
run dancetrack_w5 --steps 20 --warmup 5
run dancetrack_w15 --steps 20 --warmup 15
run mot17ckpt_w3 --config mot17 --use-checkpoint --steps 8 --warmup 3
run mot17ckpt_w14 --config mot17 --use-checkpoint --steps 8 --warmup 14
run bf16_w4 --dtype bf16 --steps 10 --warmup 4
run bf16_w14 --dtype bf16 --steps 10 --warmup 14
