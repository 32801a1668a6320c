export MEMOTR_BENCH_NO_KERNEL_LEGS=1
run() { name=$1; dir=$2; (cd $dir && python bench.py --config bdd100k --dtype f32 --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$name', round(d['ms_per_step'],2), round(d['value'],2), round(d['host_ms_per_step'],1))"); }
run head .
run old _bisect
run head .
run old _bisect
