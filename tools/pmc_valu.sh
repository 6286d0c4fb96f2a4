#!/bin/bash
# VALU activity of the two te_step kernels (one rocprofv3 --pmc pass per counter, no trace domains) -> gpurun_out/pmc_valu/pmc_valu_latest.json
# (copy to profiles/pmc_valu_latest.json: bench.py reads it for roofline.valu_*).  usage: bash tools/pmc_valu.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_valu; mkdir -p $OUT
for c in "SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE" "SQ_WAVES"; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/$c -o b -- python3 bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-profile-events --headline-only > /dev/null 2>&1 || echo "pass $c failed"
done
# the armed-drone census of the same window, from an unprofiled run of the same command
ARMED=$(python3 bench.py --steps 20 --warmup 10 --no-cpu-baseline --headline-only 2>/dev/null | python3 -c 'import json,sys; print(json.loads(sys.stdin.read())["config"]["armed_drones_per_env"])')
python3 tools/pmc_valu.py $OUT $ARMED 65536
