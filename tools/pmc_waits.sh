#!/bin/bash
# Where do the waves of the two te_step kernels spend their cycles?  One counter per --pmc pass (no trace domains):
# wave-cycles resident, waiting for an instruction slot, waiting at s_waitcnt, issuing VALU / VMEM / SALU / LDS.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_waits; mkdir -p $OUT
for c in "SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY" "SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_VMEM" "SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM" "SQ_INSTS_VMEM" "SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/$c -o b -- python3 bench.py --steps 20 --warmup ${PMC_WARMUP:-10} --no-cpu-baseline --no-profile-events --headline-only > /dev/null 2>&1 || { echo "pass $c failed"; continue; }
  f=$(find $OUT/$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 tools/pmc_summary.py $f | grep -A2 "substeps_kernel\|engage_observe" | grep -v "^--" || true
done
