# per-kernel averages (rocprofv3 --kernel-trace --stats) of a level5-family rollout, dense and with the persistent observation, after the parity
# tests of the slot-wave engage kernels.  usage: bash tools/level5_kernels.sh [tasks...]   (default: level5 level5_c1 level5_fusion level5_dumb level5_2bt)
tasks=${@:-"level5 level5_c1 level5_fusion level5_dumb level5_2bt"}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/level5_kernels; mkdir -p $O
python -m pytest tests/test_gpu_engage_slots.py tests/test_gpu_level5.py -q -x 2>&1 | tail -2 &&
cd /tmp && export TMPDIR=/tmp && for t in $tasks; do for f in "" "--persistent-obs"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$t$f -o l5 -- python3 $R/bench.py --task $t --steps 50 --warmup 20 --no-cpu-baseline --headline-only --no-profile-events $f > $O/bench_$t$f.json 2>/dev/null
  echo "$t $f: $(python3 $R/tools/bench_line.py < $O/bench_$t$f.json)"
  python3 - "$O/prof_$t$f" <<'PY'
import csv, sys, glob
for r in csv.DictReader(open(glob.glob(sys.argv[1] + "/*kernel_stats.csv")[0])):
    if any(k in r["Name"] for k in ("engage", "ring_push", "stack_view", "substeps")): print("    %-64s %8.1f us x %s" % (r["Name"][:64], float(r["AverageNs"]) / 1e3, r["Calls"]))
PY
done; done | tee $O/summary.txt
