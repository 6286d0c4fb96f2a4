#!/bin/bash
# Round profile set on the GPU box (run from the repo root through gpurun): bash tools/collect_profiles.sh <tag>
#   bench_driver_cmd.json            the driver's exact command, not profiled
#   kernel_stats_driver_cmd.csv      rocprofv3 --kernel-trace --stats of the SAME command (all regimes: the average mixes them)
#   headline_from_trace.json         the 20 launches of the timed window out of that trace (tools/headline_from_trace.py)
#   kernel_stats_headline_only.csv   rocprofv3 --stats of `--headline-only --no-profile-events`: warm-up + timed window only
#   pmc_latest.json                  HBM bytes per launch (separate --pmc passes, tools/pmc_traffic.sh)
#   pmc_valu_latest.json             VALU instructions / busy cycles per launch and per armed-drone sub-step (tools/pmc_valu.sh)
#   kernel_stats_8192_envs.csv       rocprofv3 --stats of the metric's 8-GPU shard (8 192 envs): engage_slots_kernel
set -e
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench.err
echo "bench done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o drv -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_under_rocprof.json 2>/dev/null
echo "rocprof driver cmd done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o head -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --headline-only --no-profile-events --no-cpu-baseline > $O/bench_headline_only_under_rocprof.json 2>/dev/null
echo "rocprof headline-only done"
cd $R
cp $O/prof/drv_kernel_stats.csv $O/kernel_stats_driver_cmd.csv
cp $O/prof/head_kernel_stats.csv $O/kernel_stats_headline_only.csv
python3 tools/headline_from_trace.py $O/prof/drv_kernel_trace.csv --bench-json $O/bench_driver_cmd.json > $O/headline_from_trace.json
bash tools/pmc_traffic.sh > $O/pmc.log 2>&1 || true
cp gpurun_out/pmc_traffic/pmc_latest.json $O/ 2>/dev/null || true
bash tools/pmc_valu.sh > $O/pmc_valu.log 2>&1 || true
cp gpurun_out/pmc_valu/pmc_valu_latest.json $O/ 2>/dev/null || true
# the small-shard engage kernel (engage_slots_kernel) under the profiler too: the 8-GPU shard of the metric
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o shard -- python3 $R/bench.py --gpus 1 --steps 200 --warmup 30 --envs-per-gpu 8192 --headline-only --no-profile-events --no-cpu-baseline > $O/bench_8192_under_rocprof.json 2>/dev/null
cd $R
cp $O/prof/shard_kernel_stats.csv $O/kernel_stats_8192_envs.csv 2>/dev/null || true
echo "pmc done"
python3 tools/bench_line.py < $O/bench_driver_cmd.json
cat $O/headline_from_trace.json; tail -5 $O/pmc.log; head -4 $O/kernel_stats_headline_only.csv | cut -c1-170
