# sweep forced occupancy of the sub-step kernel on the GPU box (rebuilds with hipcc there)
for w in 0 4 5; do
  flags="[]"; [ "$w" != "0" ] && flags="['-DTE_K1_WAVES=$w']"
  python -c "
from dronechase_amd.build import build_library
build_library(force=True, extra_flags=$flags)" >/dev/null 2>&1
  echo "waves_per_eu=$w: $(python tools/k2_probe.py 2>/dev/null | sed -n 2p) | N=262144: $(python tools/k2_probe.py 262144 2>/dev/null | sed -n 2p)"
done
