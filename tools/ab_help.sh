# A/B of the noise-helper waves of the sub-step kernel inside ONE gpurun call: TE_K1_HELP=0 vs 1 (te_create's default admits them on shards of
# up to kHelpMaxPairs (env, slot) pairs), interleaved, per task and shard size.
# usage: bash tools/ab_help.sh "<task flags>" sizes...
task=${1:-stage03}; shift
sizes=${@:-"4096 8192 16384 32768"}
row() { python bench.py --steps 200 --warmup 30 --no-cpu-baseline --headline-only "$@" 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline_env_step"]; print("%s x %d: %.0f M  %.1f us/step  sub-steps %.1f  engage %.1f" % (d["config"]["task"], d["config"]["envs_per_gpu"], d["value"]/1e6, d["ms_per_step"]*1e3, r["substeps_kernel_ms"]*1e3, r["engage_observe_kernel_ms"]*1e3))'; }
for n in $sizes; do for rep in 1 2; do for m in 0 1; do
  echo "help=$m  $(TE_K1_HELP=$m row --task $task --envs-per-gpu $n)"
done; done; done
