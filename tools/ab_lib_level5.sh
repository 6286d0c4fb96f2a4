# A/B of two builds of the library (ab/libA.so vs ab/libB.so) on the level5 family, interleaved: bash tools/ab_lib_level5.sh [task] [sizes...]
task=${1:-level5}; shift
sizes=${@:-"8192 65536"}
for n in $sizes; do for rep in 1 2; do for v in A B; do
  cp ab/lib$v.so dronechase_amd/libthreatengage.so
  for f in "" "--persistent-obs"; do echo "$v  $task x $n $f: $(python bench.py --task $task --envs-per-gpu $n --steps 100 --warmup 20 --no-cpu-baseline --headline-only $f 2>/dev/null | python tools/bench_line.py)"; done
done; done; done
cp ab/libB.so dronechase_amd/libthreatengage.so
