# level5 family: tests, dense / persistent table, per-kernel averages of the level5 rollout
mkdir -p gpurun_out/r04_l5
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_level5.py tests/test_gpu_fixtures.py -q -x 2>&1 | tail -3 &&
for t in level5 level5_c1 level5_fusion level5_dumb level5_2bt; do
  for f in "" "--persistent-obs"; do
    echo "$t $f: $(python bench.py --task $t --steps 100 --warmup 20 --no-cpu-baseline --headline-only $f 2>/dev/null | python tools/bench_line.py)"
  done
done | tee gpurun_out/r04_l5/table.txt &&
cd /tmp && export TMPDIR=/tmp && for f in "" "--persistent-obs"; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_l5/prof$f -o l5 -- python3 $R/bench.py --task level5 --steps 50 --warmup 20 --no-cpu-baseline --headline-only --no-profile-events $f > /dev/null 2>&1
head -6 $R/gpurun_out/r04_l5/prof$f/*kernel_stats.csv | cut -c1-160
done
