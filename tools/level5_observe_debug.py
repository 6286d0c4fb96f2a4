#!/usr/bin/env python3
"""debug aid: where does te_observe_stacked differ from the observation te_step_stacked just wrote?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dronechase_amd import default_config
from dronechase_amd.batched_env import BatchedEnv
from oracle import te_oracle as O
N = 256
cfg = default_config("level5", n_envs=N, motor_noise=1, seed=5)
g = BatchedEnv(cfg, "cuda:0"); o = O.OracleEnv(cfg, "f32")
g.reset(); o.reset()
for t in range(14):
    a = o.random_actions(9, t)
    gs, gm, *_ = g.step_stacked(torch.from_numpy(a).cuda())
step_obs, step_mask = gs.clone(), gm.clone()
done = g.done.cpu().numpy() != 0
gs2, gm2, *_ = g.observe_stacked()
torch.cuda.synchronize()
keep = torch.from_numpy(~done).cuda()
print("mask equal:", torch.equal(gm2[keep], step_mask[keep]))
d = (gs2 != step_obs) & keep.view(-1, 1, 1, 1, 1)
idx = d.nonzero()
print("differing elements:", len(idx), "in envs", sorted(set(idx[:, 0].tolist()))[:20])
for i in idx[:12]:
    t = tuple(i.tolist())
    print("  ", t, float(gs2[t]), float(step_obs[t]))
