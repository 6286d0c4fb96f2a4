# A/B of ab/libA.so vs ab/libB.so at small shards (the latency-bound BASELINE configs), interleaved
for r in 1 2; do for v in A B; do
  cp ab/lib$v.so dronechase_amd/libthreatengage.so
  for cfgl in "stage03 8192 0" "stage01 4096 0" "stage02 16384 8"; do
    set -- $cfgl
    x=""; [ $3 != 0 ] && x="--n-invaders $3"
    echo "$v $1 x $2: $(python bench.py --task $1 $x --envs-per-gpu $2 --steps 300 --warmup 30 --no-cpu-baseline --headline-only 2>/dev/null | python tools/bench_line.py)"
  done
done; done
cp ab/libB.so dronechase_amd/libthreatengage.so
