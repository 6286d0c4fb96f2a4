python -c "
from dronechase_amd.build import build_library
build_library(force=True, extra_flags=['-DTE_DEBUG_STAMPS=1'])" >/dev/null 2>&1
python tools/k2_stamps.py; python tools/k2_stamps.py 4096
