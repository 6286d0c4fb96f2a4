# the level5 engage kernel alone: parity tests of the slot-wave forms, then per-kernel averages of a level5 rollout (dense and persistent)
mkdir -p gpurun_out/r04_l5
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_engage_slots.py -q -x -k stacked 2>&1 | tail -2 &&
cd /tmp && export TMPDIR=/tmp && for t in level5 level5_c1; do for f in "" "--persistent-obs"; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_l5/prof_$t$f -o l5 -- python3 $R/bench.py --task $t --steps 50 --warmup 20 --no-cpu-baseline --headline-only --no-profile-events $f > $R/gpurun_out/r04_l5/bench_$t$f.json 2>/dev/null
echo "$t $f: $(python3 $R/tools/bench_line.py < $R/gpurun_out/r04_l5/bench_$t$f.json)"
grep -i "engage\|ring_push\|stack_view\|substeps" $R/gpurun_out/r04_l5/prof_$t$f/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
done; done
