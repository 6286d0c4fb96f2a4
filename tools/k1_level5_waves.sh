# per-wave timeline of the sub-step launch for level5 (1.6 GB background next to ~8 k flights): stamp build, three fill-wave counts
mkdir -p gpurun_out/r04_l5
python -c "from dronechase_amd.build import build_library; build_library(force=True, extra_flags=['-DTE_DEBUG_STAMPS', '-DTE_NO_LSTAMP'])" > /dev/null 2>&1 || exit 1
for w in 512 2048; do
  echo "== TE_FILL_WAVES=$w"; TE_FILL_WAVES=$w python tools/k1_waves.py 65536 60 level5
done 2>&1 | tee gpurun_out/r04_l5/k1_waves_level5.txt
