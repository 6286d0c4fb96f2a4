#!/usr/bin/env python3
"""Level5DumbMultiObs in batched form: te_step_students (seven scripted wingmen, 30 invader slots, every wingman's stacked observation per step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dronechase_amd import default_config
from dronechase_amd.batched_env import BatchedEnv
args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(args[0]) if len(args) > 0 else 16384
steps = int(args[1]) if len(args) > 1 else 100
env = BatchedEnv(default_config("level5_dumb", n_envs=N), "cuda:0")
if "--persistent-obs" in sys.argv:   # the observations are updated in place (te_set_persistent_obs): the same buffer every step
    env.set_persistent_obs(True)
env.reset()
for _ in range(20):
    env.step_students()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    out = env.step_students()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
active = float(out[4].float().sum(1).mean())
print(f"te_step_students, {N} envs: {dt * 1e3:.2f} ms/step = {N / dt / 1e6:.2f} M env-steps/s = {N * active / dt / 1e6:.1f} M student observations/s "
      f"({active:.2f} armed wingmen per env, {N * 7 * 24336 / 1e9:.2f} GB of stacked observation per step)")
env.close()
