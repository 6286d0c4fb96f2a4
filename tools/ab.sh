for cfg in "stage01 4096 0" "stage01 65536 0" "stage02 16384 8" "stage02 65536 8" "exp02 65536 0" "stage03 8192 0" "stage03 16384 0" "stage03 32768 0" "stage03 131072 0"; do
  set -- $cfg
  extra=""; [ "$3" != "0" ] && extra="--n-invaders $3"
  echo "$1 N=$2 $extra: $(python bench.py --task $1 --envs-per-gpu $2 $extra --steps 200 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline_env_step"]; print(round(d["value"]/1e6,1), "M env-steps/s,", round(d["ms_per_step"]*1e3,1), "us/step, K1", round(r["substeps_kernel_ms"]*1e3,1), "K2", round(r["engage_observe_kernel_ms"]*1e3,1), "armed/env", round(d["roofline"]["armed_drones_per_env"],2), "step frac", round(r["frac"],3))')"
done
