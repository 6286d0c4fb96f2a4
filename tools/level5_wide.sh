mkdir -p gpurun_out/r04_l5
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && for t in level5_fusion level5_dumb level5_2bt; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_l5/prof_$t -o l5 -- python3 $R/bench.py --task $t --steps 50 --warmup 20 --no-cpu-baseline --headline-only --no-profile-events > $R/gpurun_out/r04_l5/bench_$t.json 2>/dev/null
echo "$t: $(python3 $R/tools/bench_line.py < $R/gpurun_out/r04_l5/bench_$t.json)"
done
