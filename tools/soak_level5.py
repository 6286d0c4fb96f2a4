#!/usr/bin/env python3
"""1 500 random-action steps of the level5 family (te_step_stacked: slot-wave engage, ring push, stack view): stacked spheres and inertial rows
finite and in range, episodes ending and restarting, the validity mask populated.    python tools/soak_level5.py"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from dronechase_amd import default_config
from dronechase_amd.batched_env import BatchedEnv
for task, N in (("level5", 16384), ("level5_fusion", 8192), ("level5_dumb", 4096), ("level5_c1", 16384)):
    env = BatchedEnv(default_config(task, n_envs=N, seed=3), "cuda:0")
    env.reset()
    a = torch.empty((N, 4), device="cuda:0")
    dones = 0; bad = 0
    for t in range(1500):
        env.random_actions(5, t, out=a)
        st, mask, inert, last, rew, done, info = env.step_stacked(a)
        dones += int(done.sum())
        if t % 250 == 249:
            ok = bool(torch.isfinite(st).all()) and bool(torch.isfinite(inert).all()) and bool(torch.isfinite(rew).all()) and float(st.min()) >= 0.0 and float(st.max()) <= 1.0 and float(inert.abs().max()) <= 1.0
            w = env.get_state()
            bad += 0 if ok else 1
            print(f"{task} x {N} step {t + 1}: finite/in range {ok}, episodes ended {dones}, valid spheres/env {float(mask.float().sum(1).mean()):.2f}, mean reward {float(rew.mean()):.3f}", flush=True)
    env.close()
    assert bad == 0
print("soak ok")
