#!/usr/bin/env python3
"""Per-workgroup phase times of the engage/observe kernel over a rollout (needs a -DTE_DEBUG_STAMPS -DTE_LDS_STAMPS build and TE_ENGAGE=lds; that combination does not compile with ROCm 7.2 since round 3, see te_device.hpp):
which phase, in which blocks, sets the kernel's critical path."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, numpy as np, torch
from dronechase_amd import default_config
from dronechase_amd.batched_env import BatchedEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 230
env = BatchedEnv(default_config("stage03", n_envs=N), "cuda:0")
a = torch.empty((N, 4), device="cuda:0")
env.reset()
nb = N // 64
names = ["stage", "precompute", "logic", "barrier", "rows/prep", "tail"]
lnames = ["shoot", "explode+origin", "reward", "terminate", "info+hits", "round", "reset", "->end"]
n = 64 + 16 * nb
for i in range(steps):
    env.random_actions(1234, i, out=a); env.step(a)
    if i % 30 == 20 or i == steps - 1:
        out = (C.c_uint64 * n)()
        env.L.te_debug_stamps(env._h, out, n)
        t = np.frombuffer(out, dtype=np.uint64)[64:].reshape(nb, 16).astype(np.float64) * 0.01  # us
        t0 = t[:, 0].min()
        ph = np.diff(t[:, :7], axis=1)
        start = t[:, 0] - t0; end = t[:, 6] - t0
        done = env.done.view(nb, 64).any(1).cpu().numpy()
        print(f"step {i}: kernel span {end.max():.1f} us; block start p50 {np.median(start):.1f} max {start.max():.1f}; "
              f"block duration p50 {np.median(end - start):.1f} p99 {np.percentile(end - start, 99):.1f} max {(end - start).max():.1f}; "
              f"blocks with a done env {done.mean():.2f}")
        for k, nm in enumerate(names):
            print(f"    {nm:10s} p50 {np.median(ph[:, k]):6.2f}  p99 {np.percentile(ph[:, k], 99):6.2f}  max {ph[:, k].max():6.2f}   "
                  f"done-blocks p50 {np.median(ph[done, k]) if done.any() else 0:6.2f}")
        lt = np.concatenate([t[:, 2:3], t[:, 8:15], t[:, 3:4]], axis=1)  # logic start, 7 inner stamps, logic end
        lp = np.diff(lt, axis=1)
        for k, nm in enumerate(lnames):
            print(f"      logic/{nm:15s} p50 {np.median(lp[:, k]):6.2f}  p99 {np.percentile(lp[:, k], 99):6.2f}  max {lp[:, k].max():6.2f}")
