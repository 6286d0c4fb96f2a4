# A/B of the slots-per-wave choice of the engage kernel inside ONE gpurun call: TE_SLOT_SPW=1 (engage_slots_kernel: one wave per slot) vs
# TE_SLOT_SPW=2 (engage_slots_multi_kernel<2, true>), interleaved, all three regimes of bench.py.  usage: bash tools/ab_spw.sh [task] [sizes...]
task=${1:-stage03}; shift
sizes=${@:-"32768 65536"}
row() { python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-profile-events "$@" 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); c=d["config"]; print("%s x %d: headline %.0f M  %.1f us/step; steady %.0f M; all-armed %.0f M" % (c["task"], c["envs_per_gpu"], d["value"]/1e6, d["ms_per_step"]*1e3, d.get("steady_state",{}).get("value",0)/1e6, d.get("all_armed",{}).get("value",0)/1e6))'; }
for n in $sizes; do for rep in 1 2; do for m in 1 2; do
  echo "spw=$m  $(TE_SLOT_SPW=$m row --task $task --envs-per-gpu $n)"
done; done; done
