# null stream vs a stream of the bench's own, interleaved inside one gpurun call
for r in 1 2 3; do
  for w in "--steps 20 --warmup 5" "--steps 100 --warmup 20"; do
    echo "null  [$w]: $(python bench.py $w --no-cpu-baseline --headline-only 2>/dev/null | python tools/bench_line.py)"
    echo "side  [$w]: $(python bench.py $w --no-cpu-baseline --headline-only --own-stream 2>/dev/null | python tools/bench_line.py)"
  done
done
