# A/B of two builds of the library inside ONE gpurun call (ab/libA.so vs ab/libB.so), interleaved, all three regimes of bench.py.
# usage: bash tools/ab_lib_regimes.sh "<task flags>" sizes...
task=${1:-stage03}; shift
sizes=${@:-"65536"}
row() { python bench.py --steps 200 --warmup 30 --no-cpu-baseline "$@" 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); c=d["config"]; r=d["roofline_env_step"]; s=d.get("steady_state",{}); a=d.get("all_armed",{}); print("%s x %d: headline %.0f M %.1f us (K1 %.1f K2 %.1f); steady %.0f M; all-armed %.0f M" % (c["task"], c["envs_per_gpu"], d["value"]/1e6, d["ms_per_step"]*1e3, r["substeps_kernel_ms"]*1e3, r["engage_observe_kernel_ms"]*1e3, s.get("value",0)/1e6, a.get("value",0)/1e6))'; }
for n in $sizes; do for rep in 1 2; do for v in A B; do
  cp ab/lib$v.so dronechase_amd/libthreatengage.so
  echo "$v  $(row --task $task --envs-per-gpu $n)"
done; done; done
cp ab/libB.so dronechase_amd/libthreatengage.so
