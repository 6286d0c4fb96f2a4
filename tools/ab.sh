for r in 1 2 3; do for v in A B; do
  cp ab/lib$v.so dronechase_amd/libthreatengage.so
  echo "$v: $(python bench.py --steps 200 --warmup 30 --no-cpu-baseline --headline-only 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"]/1e6,1), round(d["roofline_env_step"]["substeps_kernel_ms"]*1e3,1), round(d["roofline_env_step"]["engage_observe_kernel_ms"]*1e3,1))')"
done; done
cp ab/libA.so dronechase_amd/libthreatengage.so
