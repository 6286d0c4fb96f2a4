#!/bin/bash
# Collect HBM traffic of the two te_step kernels with rocprofv3 PMC counters (separate passes for FETCH_SIZE and
# WRITE_SIZE, plus the calibration program), on the GPU box.  Outputs under gpurun_out/pmc_traffic/.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_traffic; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -o $OUT/pmc_calib tools/pmc_calib.hip
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/calib_$c -o c -- $OUT/pmc_calib > /dev/null 2>&1
  rocprofv3 --pmc $c --output-format csv -d $OUT/bench_$c -o b -- python3 bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-profile-events --headline-only > /dev/null 2>&1
done
python3 tools/pmc_traffic.py $OUT
