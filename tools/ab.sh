# A/B of two builds of the library inside ONE gpurun call (box-to-box spread is +-3 %): ab/libA.so vs ab/libB.so, three interleaved
# repetitions of the full bench (headline, steady state, all armed).  usage: bash tools/ab.sh [extra bench.py flags]
for r in 1 2 3; do for v in A B; do
  cp ab/lib$v.so dronechase_amd/libthreatengage.so
  echo "$v: $(python bench.py --steps 100 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | python tools/bench_line.py)"
done; done
cp ab/libB.so dronechase_amd/libthreatengage.so
