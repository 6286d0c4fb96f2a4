for w in 0 5; do
  flags="[]"; [ "$w" != "0" ] && flags="['-DTE_K1_WAVES=$w']"
  python -c "
from dronechase_amd.build import build_library
build_library(force=True, extra_flags=$flags)" >/dev/null 2>&1
  echo "waves_per_eu=$w: $(python tools/k2_probe.py 2>/dev/null | sed -n 1p)"
done
