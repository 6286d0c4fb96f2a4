# dense vs persistent observation (te_set_persistent_obs), one box
for t in stage03 stage02 stage01 level5 level5_c1 level5_fusion; do
  x=""; [ $t = stage02 ] && x="--n-invaders 8"
  for f in "" "--persistent-obs"; do
    echo "$t $f: $(python bench.py --task $t $x --steps 100 --warmup 20 --no-cpu-baseline $f 2>/dev/null | python tools/bench_line.py)"
  done
done
python tools/students_bench.py 65536 30 2>/dev/null
python tools/students_bench.py 65536 30 --persistent-obs 2>/dev/null
