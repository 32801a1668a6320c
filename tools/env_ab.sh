mkdir -p gpurun_out/r06z
run() { name=$1; shift; env "$@" python bench.py --workload train --steps 10 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$name', round(d['ms_per_step'],2), round(d['value'],2), round(d['host_ms_per_step'],1))"; }
export MEMOTR_BENCH_NO_KERNEL_LEGS=1
run base A=1
run kernarg1 HIP_FORCE_DEV_KERNARG=1
run kernarg0 HIP_FORCE_DEV_KERNARG=0
run base2 A=1
run hwq2 GPU_MAX_HW_QUEUES=2
run nosdma HSA_ENABLE_SDMA=0
run activewait ROC_ACTIVE_WAIT_TIMEOUT=1000
run base3 A=1
