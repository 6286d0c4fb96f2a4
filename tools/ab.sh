# A/B of two prebuilt libraries (ab/libA.so = HEAD, ab/libB.so = working tree) inside one gpurun call
cp ab/libB.so dronechase_amd/libthreatengage.so
python -m pytest tests -m gpu -x -q 2>&1 | tail -1
for r in 1 2 3; do for v in A B; do
  cp ab/lib$v.so dronechase_amd/libthreatengage.so
  echo "$v: $(python bench.py --steps 200 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"]/1e6,1), round(d["roofline_env_step"]["substeps_kernel_ms"]*1e3,1), round(d["roofline_env_step"]["engage_observe_kernel_ms"]*1e3,1))')"
done; done
cp ab/libB.so dronechase_amd/libthreatengage.so
