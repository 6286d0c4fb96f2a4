#!/usr/bin/env python3
"""Per-wave phase times of engage_slots_kernel (te_engage_slots.hpp): waves 0 (agent) and 1 (ally) of every workgroup.  Needs a stamp build:
    python -c "from dronechase_amd.build import build_library; build_library(force=True, extra_flags=['-DTE_DEBUG_STAMPS', '-DTE_NO_LSTAMP', '-DTE_NO_ESTAMP'])"
    TE_ENGAGE=slots python tools/slots_stamps.py [N] [steps] [task]     (level5 / level5_c1: engage_slots_stacked_kernel)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, numpy as np, torch
from dronechase_amd import default_config
from dronechase_amd.batched_env import BatchedEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 320
task = sys.argv[3] if len(sys.argv) > 3 else "stage03"
cfg = default_config(task, n_envs=N)
env = BatchedEnv(cfg, "cuda:0")
a = torch.empty((N, 4), device="cuda:0")
env.reset()
nb = N // 64
names = ["loads", "own slot + targeting", "barrier 1", "engagement + patches + spawn", "barrier 2", "rows / commands / plan", "drain stores"]
n = 64 + 16 * nb
for i in range(steps):
    env.random_actions(1234, i, out=a)
    if cfg.stacked_obs: env.step_stacked(a)
    else: env.step(a)
    if i in (20, 100, 300, steps - 1):
        torch.cuda.synchronize()
        out = (C.c_uint64 * n)()
        env.L.te_debug_stamps(env._h, out, n)
        t = np.frombuffer(out, dtype=np.uint64)[64:].reshape(nb, 2, 8).astype(np.float64) * 0.01  # 100 MHz -> us
        t0 = t[:, :, 0].min()
        for w in (0, 1):
            ph = np.diff(t[:, w, :], axis=1)
            start, end = t[:, w, 0] - t0, t[:, w, 7] - t0
            print(f"step {i} wave {w}: span {end.max():.1f} us; start p50 {np.median(start):.1f} max {start.max():.1f}; "
                  f"duration p50 {np.median(end - start):.1f} p99 {np.percentile(end - start, 99):.1f} max {(end - start).max():.1f}")
            for k, nm in enumerate(names):
                print(f"    {nm:30s} p50 {np.median(ph[:, k]):6.2f}  p99 {np.percentile(ph[:, k], 99):6.2f}  max {ph[:, k].max():6.2f}")
