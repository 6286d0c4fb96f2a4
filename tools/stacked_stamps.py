#!/usr/bin/env python3
"""Per-workgroup phase times of stacked_kernel (te_stacked.hpp) over a level5 rollout.  Needs a stamp build:
    python -c "from dronechase_amd.build import build_library; build_library(force=True, extra_flags=['-DTE_DEBUG_STAMPS', '-DTE_NO_LSTAMP'])"
    python tools/stacked_stamps.py [N] [steps] [task]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, numpy as np, torch
from dronechase_amd import default_config
from dronechase_amd.batched_env import BatchedEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
task = sys.argv[3] if len(sys.argv) > 3 else "level5"
env = BatchedEnv(default_config(task, n_envs=N), "cuda:0")
a = torch.empty((N, 4), device="cuda:0")
env.reset()
nb = N // 64
names = ["stage snapshot + quaternions", "(1) own spheres of the wingmen + ring push", "(2) draws", "(3) neighbour re-projection", "(4ab) stack order, mask, terminal tiles",
         "(4c) patches", "(4d) ring reset"]
n = 64 + 16 * nb
for i in range(steps):
    env.random_actions(1234, i, out=a); env.step_stacked(a)
    if i in (20, steps // 2, steps - 1):
        torch.cuda.synchronize()
        out = (C.c_uint64 * n)()
        env.L.te_debug_stamps(env._h, out, n)
        t = np.frombuffer(out, dtype=np.uint64)[64:].reshape(nb, 16).astype(np.float64) * 0.01
        t0 = t[:, 0].min()
        ph = np.diff(t[:, :8], axis=1)
        start, end = t[:, 0] - t0, t[:, 7] - t0
        print(f"step {i}: span {end.max():.1f} us; block start p50 {np.median(start):.1f} p90 {np.percentile(start, 90):.1f} max {start.max():.1f}; "
              f"block duration p50 {np.median(end - start):.1f} p99 {np.percentile(end - start, 99):.1f}")
        for k, nm in enumerate(names):
            print(f"    {nm:44s} p50 {np.median(ph[:, k]):7.2f}  p99 {np.percentile(ph[:, k], 99):7.2f}")
env.close()
