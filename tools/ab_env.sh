# A/B of one environment switch of te_create inside ONE gpurun call, interleaved, all three regimes of bench.py.
# usage: bash tools/ab_env.sh VAR A B [task] [sizes...]      e.g.  bash tools/ab_env.sh TE_SLOT_WPE8 0 1 stage03 65536 49152
var=$1; a=$2; b=$3; task=${4:-stage03}; shift 4
sizes=${@:-"65536"}
row() { python bench.py --steps 200 --warmup 30 --no-cpu-baseline "$@" 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); c=d["config"]; r=d["roofline_env_step"]; s=d.get("steady_state",{}); a=d.get("all_armed",{}); ks=s.get("kernels",{}); print("%s x %d: headline %.0f M %.1f us (K1 %.1f K2 %.1f); steady %.0f M %.1f us (K2 %.1f); all-armed %.0f M" % (c["task"], c["envs_per_gpu"], d["value"]/1e6, d["ms_per_step"]*1e3, r["substeps_kernel_ms"]*1e3, r["engage_observe_kernel_ms"]*1e3, s.get("value",0)/1e6, s.get("ms_per_step",0)*1e3, ks.get("engage_observe_kernel_ms",0)*1e3, a.get("value",0)/1e6))'; }
for n in $sizes; do for rep in 1 2; do for v in $a $b; do
  echo "$var=$v  $(env $var=$v bash -c "$(declare -f row); row --task $task --envs-per-gpu $n")"
done; done; done
