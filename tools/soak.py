#!/usr/bin/env python3
"""Long random-action rollout at full size: is every state word finite, do the observations stay in range, how long do episodes
last, how fast do the drones spin — per quadrotor table (quad_preset 1 = the default, 0 = the recalled one).
    python tools/soak.py [task] [steps] [n_envs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dronechase_amd import config as K, default_config
from dronechase_amd.batched_env import BatchedEnv
task = sys.argv[1] if len(sys.argv) > 1 else "stage03"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
N = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
for preset in (1, 0):
    cfg = default_config(task, n_envs=N, quad_preset=preset, seed=2)
    env = BatchedEnv(cfg, "cuda:0")
    env.reset()
    D = int(cfg.n_drones)
    a = torch.empty((N, 4), device="cuda:0")
    dones = torch.zeros((), device="cuda:0"); rsum = torch.zeros((), device="cuda:0", dtype=torch.float64)
    bad_obs = 0
    for t in range(steps):
        env.random_actions(77, t, out=a)
        lidar, inertial, last, rew, done, info = env.step(a)
        dones += done.sum(); rsum += rew.double().sum()
        if t % 500 == 499 or t == steps - 1:
            w = env.get_state()
            dr = w[: N * D * K.DRONE_WORDS].view(N, D, K.DRONE_WORDS)
            fl = dr[:, :, : K.D["ARMED"]].view(torch.float32)
            armed = dr[:, :, K.D["ARMED"]] != 0
            finite = bool(torch.isfinite(fl).all())
            om = fl[:, :, K.D["OMEGA"]:K.D["OMEGA"] + 3].norm(dim=-1)[armed]
            tilt = (1 - 2 * (fl[:, :, K.D["QUAT"]] ** 2 + fl[:, :, K.D["QUAT"] + 1] ** 2)).clamp(-1, 1).acos()[armed]
            in_range = bool((inertial.abs() <= 1.0 + 1e-6).all()) and bool(((lidar >= 0) & (lidar <= 1)).all())
            bad_obs += 0 if in_range else 1
            print(f"preset {preset} step {t + 1}: finite {finite}, obs in range {in_range}, armed/env {float(armed.float().sum(1).mean()):.2f}, "
                  f"|omega| p50 {float(om.median()):.2f} p99.9 {float(om.quantile(0.999)) if om.numel() < 16_000_000 else float(om[:16_000_000].quantile(0.999)):.1f} max {float(om.max()):.1f} rad/s, "
                  f"tilt p50 {float(tilt.median()):.3f} max {float(tilt.max()):.2f} rad, episodes ended {int(dones)}, mean reward/step {float(rsum) / ((t + 1) * N):.3f}", flush=True)
    print(f"preset {preset}: mean episode length {steps * N / max(int(dones), 1):.1f} steps; observation range violations {bad_obs}")
    env.close()
