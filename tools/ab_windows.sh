# A/B of ab/libA.so vs ab/libB.so over several windows of the rollout (warm-up W, then 60 timed steps)
for W in 30 100 180 300 1000; do for v in A B; do
  cp ab/lib$v.so dronechase_amd/libthreatengage.so
  echo "W=$W $v: $(python bench.py --steps 60 --warmup $W --no-cpu-baseline --headline-only 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"]/1e6,1), round(d["roofline_env_step"]["substeps_kernel_ms"]*1e3,1), round(d["roofline_env_step"]["engage_observe_kernel_ms"]*1e3,1), round(d["roofline"]["armed_drones_per_env"],2))')"
done; done
cp ab/libA.so dronechase_amd/libthreatengage.so
