# A/B of sub-step kernel variants inside ONE gpurun call (box-to-box timing varies by several percent)
for f in "" "-DTE_K1_SPREAD=1" "-DTE_K1_REMAP=1" "-DTE_K1_SPREAD=1 -DTE_K1_REMAP=1"; do
  python -c "
from dronechase_amd.build import build_library
build_library(force=True, extra_flags='$f'.split())" >/dev/null 2>&1
  python -m pytest tests/test_gpu_properties.py -m gpu -q -x 2>&1 | tail -1
  echo "[$f]: $(python tools/k2_probe.py 2>/dev/null | sed -n 1p); bench: $(python bench.py --steps 200 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"]/1e6,1), round(d["roofline_env_step"]["substeps_kernel_ms"]*1e3,1), round(d["roofline_env_step"]["engage_observe_kernel_ms"]*1e3,1))')"
done
