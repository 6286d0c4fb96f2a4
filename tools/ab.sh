cd /tmp && export TMPDIR=/tmp
for lim in 1 2 3 0; do
  rm -rf /tmp/p5
  TE_STACK_LIMIT=$lim rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p5 -o l5 -- python3 $GRAFT_REPO_ROOT/tools/level5_bench.py 65536 40 > /dev/null 2>&1
  echo "limit $lim: $(find /tmp/p5 -name '*kernel_stats.csv' -exec grep stacked_kernel {} \; | cut -d, -f4)"
done
