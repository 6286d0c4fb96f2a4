#!/usr/bin/env python3
"""Where do engage_slots_kernel and engage_kernel differ?  (debug aid for tests/test_gpu_engage_slots.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dronechase_amd import default_config, config as K
from dronechase_amd.batched_env import BatchedEnv

task, n, steps = sys.argv[1] if len(sys.argv) > 1 else "stage03", int(sys.argv[2]) if len(sys.argv) > 2 else 8192, int(sys.argv[3]) if len(sys.argv) > 3 else 40
over = dict(kv.split("=") for kv in sys.argv[4:]); over = {k: int(v) for k, v in over.items()}
envs = []
VAR, A, B = os.environ.get("AB_VAR", "TE_ENGAGE"), os.environ.get("AB_A", "slots"), os.environ.get("AB_B", "regs")   # e.g. AB_VAR=TE_K1_HELP AB_A=1 AB_B=0
for mode in (A, B):
    os.environ[VAR] = mode
    envs.append(BatchedEnv(default_config(task, n_envs=n, **over), "cuda:0"))
a, b = envs
a.reset(); b.reset()
names = ["lidar", "inertial", "last_action", "reward", "done", "info"]
for s in range(steps):
    act = a.random_actions(11, s)
    oa, ob = a.step(act), b.step(act)
    bad = False
    for k, (x, y) in enumerate(zip(oa, ob)):
        if not torch.equal(x, y):
            bad = True
            d = (x != y)
            idx = d.nonzero()
            print(f"step {s}: {names[k]} differs in {int(d.sum())} elements of {d.numel()}; first:", idx[:5].tolist())
            for i in idx[:5]:
                t = tuple(i.tolist())
                print("   ", t, float(x[t]), float(y[t]))
    sa, sb = a.get_state(), b.get_state()
    if not torch.equal(sa, sb):
        bad = True
        d = (sa != sb).nonzero().flatten()
        D = a.cfg.n_drones
        print(f"step {s}: state differs in {len(d)} words; first:")
        for w in d[:10].tolist():
            if w < n * D * K.DRONE_WORDS:
                e, r = divmod(w, D * K.DRONE_WORDS); sl, wd = divmod(r, K.DRONE_WORDS)
                nm = [k for k, v in K.D.items() if v <= wd][-1] if hasattr(K, "D") else "?"
                print(f"    env {e} slot {sl} word {wd} ({nm}): {int(sa[w])} vs {int(sb[w])}")
            else:
                w2 = w - n * D * K.DRONE_WORDS
                e, wd = divmod(w2, K.ENV_WORDS)
                print(f"    env {e} env-word {wd}: {int(sa[w])} vs {int(sb[w])}")
    if bad:
        break
else:
    print("identical over", steps, "steps")
