set -u
mkdir -p gpurun_out/profiles gpurun_out/lines
export TMPDIR=/tmp
bash tools/traffic_probe.sh r06 > gpurun_out/r06_traffic.log 2>&1
bash tools/prof.sh r06_msda msda > gpurun_out/r06_prof_msda.log 2>&1
bash tools/prof.sh r06_train train > gpurun_out/r06_prof_train.log 2>&1
bash tools/pmc_probe.sh r06_fwd_win fwd > /dev/null 2>&1
bash tools/pmc_probe.sh r06_bwd_bins bwd > /dev/null 2>&1
cp profiles/traffic.json profiles/traffic_bwd.json /tmp/ 2>/dev/null
# the bench reads profiles/traffic*.json: use the fresh ones for the lines below
cp gpurun_out/profiles/traffic.json gpurun_out/profiles/traffic_bwd.json profiles/ 2>/dev/null
bash tools/bench_lines.sh rest r06 > gpurun_out/r06_lines.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/lines/r06_bench_default_driver_cmd.err | tail -1 > gpurun_out/lines/r06_bench_default_driver_cmd.json
cut -c1-220 gpurun_out/lines/r06_bench_default_driver_cmd.json
cat gpurun_out/r06_lines.log | tail -12
ls gpurun_out/profiles
