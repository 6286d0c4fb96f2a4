#!/bin/bash
# Round profile set on the GPU box: bench line, rocprofv3 kernel stats of the same command, PMC traffic.
# usage (from the repo root, through gpurun): bash tools/collect_profiles.sh <tag>
set -e
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
python3 $R/bench.py > $R/gpurun_out/$TAG/bench.json 2> $R/gpurun_out/$TAG/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/prof -o $TAG -- python3 $R/bench.py --no-cpu-baseline --headline-only > $R/gpurun_out/$TAG/bench_under_rocprof.json 2>/dev/null
cd $R && bash tools/pmc_traffic.sh > gpurun_out/$TAG/pmc.log 2>&1
cp gpurun_out/pmc_traffic/pmc_latest.json gpurun_out/$TAG/ 2>/dev/null || true
find gpurun_out/$TAG/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/$TAG/kernel_stats.csv \;
cat gpurun_out/$TAG/bench.json; tail -5 gpurun_out/$TAG/pmc.log; head -4 gpurun_out/$TAG/kernel_stats.csv | cut -c1-170
