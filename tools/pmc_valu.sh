#!/bin/bash
# VALU activity of the two te_step kernels (separate --pmc pass, no trace domains): instructions issued and busy cycles
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_valu; mkdir -p $OUT
for c in "SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE" "SQ_WAVES"; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/$c -o b -- python3 bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-profile-events --headline-only > /dev/null 2>&1 || echo "pass $c failed"
  f=$(find $OUT/$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 tools/pmc_summary.py $f | grep -A2 "substeps_kernel\|engage_" | grep -v "^--" || true
done
