#!/usr/bin/env python3
"""Per-wave phase times of engage_kernel (te_engage.hpp) over a rollout.  Needs a -DTE_DEBUG_STAMPS build:
    python -c "from dronechase_amd.build import build_library; build_library(force=True, extra_flags=['-DTE_DEBUG_STAMPS', '-DTE_NO_LSTAMP'])"
    python tools/engage_stamps.py [N] [steps] [task]   (TE_ENGAGE=regs for the level4 family: the slot-wave kernel has its own tool)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, numpy as np, torch
from dronechase_amd import default_config
from dronechase_amd.batched_env import BatchedEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1100
task = sys.argv[3] if len(sys.argv) > 3 else "stage03"
cfg = default_config(task, n_envs=N)
env = BatchedEnv(cfg, "cuda:0")
a = torch.empty((N, 4), device="cuda:0")
env.reset()
nb = N // 64
names = ["loads", "masks+closest", "engagement", "reward+term+out", "lidar", "snap+terminal+patch", "spawn", "obs rows", "refs+ally cmds", "plan", "drain stores"]
n = 64 + 16 * nb
for i in range(steps):
    env.random_actions(1234, i, out=a)
    if cfg.stacked_obs: env.step_stacked(a)
    else: env.step(a)
    if i in (20, 300, steps - 1):
        torch.cuda.synchronize()
        out = (C.c_uint64 * n)()
        env.L.te_debug_stamps(env._h, out, n)
        t = np.frombuffer(out, dtype=np.uint64)[64:].reshape(nb, 16).astype(np.float64) * 0.01  # 100 MHz -> us
        t0 = t[:, 0].min()
        ph = np.diff(t[:, :12], axis=1)
        start, end = t[:, 0] - t0, t[:, 11] - t0
        print(f"step {i}: span {end.max():.1f} us; wave start p50 {np.median(start):.1f} max {start.max():.1f}; "
              f"wave duration p50 {np.median(end - start):.1f} p99 {np.percentile(end - start, 99):.1f} max {(end - start).max():.1f}")
        for k, nm in enumerate(names):
            print(f"    {nm:20s} p50 {np.median(ph[:, k]):6.2f}  p99 {np.percentile(ph[:, k], 99):6.2f}  max {ph[:, k].max():6.2f}")
        sub = t[:, 12:15] - t[:, 5:6]   # inside "snap+terminal+patch": after the snapshot planes, after terminal rows / patches, after the reward
        print("    inside phase 5->6: snapshot planes done at p50 %.2f, terminal + patches at %.2f, reward at %.2f us" % tuple(np.median(sub, axis=0)))
