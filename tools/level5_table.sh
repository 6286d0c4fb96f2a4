# level5 family, dense vs persistent observation (te_set_persistent_obs), one box
mkdir -p gpurun_out/r03_e
for t in level5 level5_c1 level5_fusion; do
  for f in "" "--persistent-obs"; do
    echo "$t $f: $(python bench.py --task $t --steps 100 --warmup 20 --no-cpu-baseline --headline-only $f 2>/dev/null | python tools/bench_line.py)"
  done
done
python tools/students_bench.py 65536 30
python tools/students_bench.py 65536 30 --persistent-obs
