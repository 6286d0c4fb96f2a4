#!/usr/bin/env python3
"""Phase breakdown of one engage/observe workgroup (needs a -DTE_DEBUG_STAMPS -DTE_LDS_STAMPS build and TE_ENGAGE=lds; that combination does not compile with ROCm 7.2 since round 3, see te_device.hpp)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, numpy as np, torch
from dronechase_amd import default_config
from dronechase_amd.batched_env import BatchedEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
W = int(sys.argv[2]) if len(sys.argv) > 2 else 20      # steps before the 40 sampled ones
env = BatchedEnv(default_config("stage03", n_envs=N), "cuda:0")
a = torch.empty((N, 4), device="cuda:0")
env.reset()
names = ["stage", "precompute", "logic", "barrier", "rows", "prepare", "patch"]
acc = np.zeros(7)
for i in range(W + 40):
    env.random_actions(1, i, out=a); env.step(a)
    if i >= W:
        out = (C.c_uint64 * 8)()
        env.L.te_debug_stamps(env._h, out, 8)
        t = np.array(list(out), dtype=np.float64)
        acc += np.diff(t) * 0.01  # 100 MHz -> us
print("K2 phases (us, block 500):", {n: round(v / 40, 2) for n, v in zip(names, acc)}, "sum", round(acc.sum() / 40, 2))
