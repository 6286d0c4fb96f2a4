#!/usr/bin/env python3
"""The SB3 drop-in surface itself: ThreatEngageVecEnv.step() with numpy observations (what `collect_rollouts` sees), against the same
class with output="torch" / infos="lazy" (what dronechase_amd.ppo uses).  The numpy mode pays PCIe for the 4.1 KB/env observation and
Python for the per-env info dicts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dronechase_amd.vec_env import ThreatEngageVecEnv

for n in (4096, 65536):
    for output, infos in (("numpy", "dicts"), ("numpy", "lazy"), ("torch", "lazy")):
        env = ThreatEngageVecEnv("stage03", num_envs=n, output=output, infos=infos)
        env.reset()
        rng = np.random.default_rng(0)
        acts = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
        if output == "torch":
            acts = torch.from_numpy(acts).cuda()
        steps = 30 if output == "numpy" else 200
        for _ in range(5):
            env.step(acts)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            env.step(acts)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print(f"VecEnv.step {n:6d} envs, output={output:5s} infos={infos:5s}: {dt * 1e3:8.2f} ms/step = {n / dt / 1e6:7.2f} M env-steps/s", flush=True)
        env.close()
