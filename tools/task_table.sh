# the "other tasks and shard sizes" table of DESIGN.md 4: one bench line per row (200 steps, no CPU baseline)
row() { python bench.py --steps 200 --no-cpu-baseline --headline-only "$@" 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline_env_step"]; print("| %s | %d | %.1f | %.0f M | %.1f | %.1f | %.1f |" % (d["config"]["task"], d["config"]["envs_per_gpu"], d["roofline"]["armed_drones_per_env"], d["value"]/1e6, d["ms_per_step"]*1e3, r["substeps_kernel_ms"]*1e3, r["engage_observe_kernel_ms"]*1e3))'; }
row --task stage01 --envs-per-gpu 4096
row --task stage01 --envs-per-gpu 65536
row --task stage02 --n-invaders 8 --envs-per-gpu 16384
row --task stage02 --n-invaders 8 --envs-per-gpu 65536
row --task exp02 --envs-per-gpu 65536
row --task exp04 --envs-per-gpu 65536
row --task exp05 --envs-per-gpu 65536
row --task evaluation --envs-per-gpu 65536
for n in 8192 16384 32768 65536 131072; do row --task stage03 --envs-per-gpu $n; done
row --task level5 --envs-per-gpu 65536
row --task level5_c1 --envs-per-gpu 65536
row --task level5_fusion --envs-per-gpu 65536
row --task level5_2bt --envs-per-gpu 65536
