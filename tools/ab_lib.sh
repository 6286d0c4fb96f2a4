# A/B of two builds of the library inside ONE gpurun call (ab/libA.so vs ab/libB.so), interleaved, headline window only, per shard size.
# usage: bash tools/ab_lib.sh "<task flags>" sizes...
task=${1:-stage03}; shift
sizes=${@:-"8192 65536"}
row() { python bench.py --steps 200 --warmup 30 --no-cpu-baseline --headline-only "$@" 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline_env_step"]; print("%s x %d: %.0f M  %.1f us/step  sub-steps %.1f  engage %.1f" % (d["config"]["task"], d["config"]["envs_per_gpu"], d["value"]/1e6, d["ms_per_step"]*1e3, r["substeps_kernel_ms"]*1e3, r["engage_observe_kernel_ms"]*1e3))'; }
for n in $sizes; do for rep in 1 2; do for v in A B; do
  cp ab/lib$v.so dronechase_amd/libthreatengage.so
  echo "$v  $(row --task $task --envs-per-gpu $n)"
done; done; done
cp ab/libA.so dronechase_amd/libthreatengage.so
