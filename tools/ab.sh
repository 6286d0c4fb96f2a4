python -m pytest tests -m gpu -q -x 2>&1 | tail -1
for r in 1 2; do for w in 256 512 768; do
  echo "waves $w: $(TE_FILL_WAVES=$w python bench.py --steps 200 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"]/1e6,1), round(d["roofline_env_step"]["substeps_kernel_ms"]*1e3,1), round(d["roofline_env_step"]["engage_observe_kernel_ms"]*1e3,1))')"
done; done
