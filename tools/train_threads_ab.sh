for t in default 8 2; do
  if [ "$t" = "default" ]; then export MEMOTR_NO_QUOTA_CAP=1; unset OMP_NUM_THREADS; else unset MEMOTR_NO_QUOTA_CAP; export OMP_NUM_THREADS=$t; fi
  python bench.py --no-cpu-baseline --steps 12 --warmup 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('threads=$t', 'train fps', round(d['value'],2), 'ms/step', round(d['ms_per_step'],1))"
done
unset OMP_NUM_THREADS MEMOTR_NO_QUOTA_CAP
python bench.py --workload infer --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | tail -1 > gpurun_out/bench_infer_r04a.json; python -c "import json; d=json.load(open('gpurun_out/bench_infer_r04a.json')); print('infer fps', d['value'], d['ms_per_step'])"
