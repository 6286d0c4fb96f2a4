#!/usr/bin/env python3
"""How many (slot, chunk) flight waves does the sub-step kernel launch, and how many would it need if the sparsely armed
slots of a chunk of 64 envs shared waves?  For a random-action rollout of `task` (default stage03, 65 536 envs): per
checkpoint the mean armed drones per env, active waves per chunk today (slots with >= 1 armed env), and waves per chunk
with slots below `dense_min` armed envs compacted into mixed waves of 64 (env, slot) items."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dronechase_amd import config as K, default_config
from dronechase_amd.batched_env import BatchedEnv

task = sys.argv[1] if len(sys.argv) > 1 else "stage03"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dense_min = int(sys.argv[3]) if len(sys.argv) > 3 else 32
env = BatchedEnv(default_config(task, n_envs=N, seed=0), "cuda:0")
env.reset()
D = env.D
a = torch.empty((N, 4), device="cuda:0")
marks = [0, 30, 100, 180, 330, 1000, 3000, 10000]
t = 0
for m in marks:
    while t < m:
        env.random_actions(12345, t, out=a); env.step(a, terminal=False); t += 1
    w = env.get_state()
    armed = (w[: N * D * K.DRONE_WORDS].view(N, D, K.DRONE_WORDS)[:, :, K.D["ARMED"]] != 0)
    per_chunk = armed.view(N // 64, 64, D).sum(1)                      # [chunks, D] armed envs per (chunk, slot)
    active = (per_chunk > 0).sum(1).float()
    dense = (per_chunk >= dense_min)
    sparse_items = (per_chunk * (~dense)).sum(1)
    compact = dense.sum(1).float() + torch.ceil(sparse_items.float() / 64)
    lanes = armed.float().sum(1).mean().item()
    print(f"step {t:5d}: armed/env {lanes:5.2f}   active waves/chunk {active.mean().item():5.2f} (max {int(active.max())})   "
          f"compacted {compact.mean().item():5.2f} (max {int(compact.max())})   lane use {lanes / active.mean().item():.2f} -> {lanes / compact.mean().item():.2f}")
