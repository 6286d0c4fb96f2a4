for stop in 1 2 3 4 5 0; do
  python -c "
from dronechase_amd.build import build_library
build_library(force=True, extra_flags=['-DTE_K2_STOP=$stop'])" >/dev/null 2>&1
  echo "STOP=$stop: $(python tools/k2_probe.py 2>/dev/null | sed -n 2p)"
done
