python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -m gpu -q -x 2>&1 | tail -2
for cfg in "256 0" "256 45" "512 0" "1024 0"; do
  set -- $cfg
  echo "[fill waves $1 pos $2%]"; TE_FILL_WAVES=$1 TE_FILL_POS=$2 python tools/k1_phase.py 65536 230 2>/dev/null | awk 'NR%3==1' | cut -c1-48
done
python bench.py --steps 200 --warmup 30 --no-cpu-baseline
