# A/B of the two engage kernels inside ONE gpurun call: TE_ENGAGE=regs (engage_kernel: one wave per chunk) vs TE_ENGAGE=slots
# (engage_slots_kernel: one wave per (chunk, slot)), interleaved, per shard size.  usage: bash tools/ab_engage.sh [task] [sizes...]
task=${1:-stage03}; shift
sizes=${@:-"4096 8192 16384 32768 65536"}
row() { python bench.py --steps 200 --warmup 30 --no-cpu-baseline --headline-only "$@" 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline_env_step"]; print("%s x %d: %.0f M  %.1f us/step  sub-steps %.1f  engage %.1f" % (d["config"]["task"], d["config"]["envs_per_gpu"], d["value"]/1e6, d["ms_per_step"]*1e3, r["substeps_kernel_ms"]*1e3, r["engage_observe_kernel_ms"]*1e3))'; }
for n in $sizes; do for rep in 1 2; do for m in regs slots; do
  echo "$m  $(TE_ENGAGE=$m row --task $task --envs-per-gpu $n)"
done; done; done
