import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dronechase_amd import default_config
from dronechase_amd.batched_env import BatchedEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cfg = default_config("level5", n_envs=N, seed=1)
def run():
    g = BatchedEnv(cfg, "cuda:0"); g.reset()
    a = torch.empty((N, 4), device="cuda:0")
    outs = []
    for t in range(12):
        g.random_actions(5, t, out=a)
        s, m, *_ = g.step_stacked(a, terminal=False)
        torch.cuda.synchronize()
        outs.append((s.clone(), m.clone()))
    return outs
A, B = run(), run()
for t, ((s, m), (s2, m2)) in enumerate(zip(A, B)):
    de = (s != s2).reshape(N, -1).any(1)
    print(f"step {t}: mask equal {torch.equal(m, m2)}  envs differing {int(de.sum())}", end="")
    if de.any():
        e = int(torch.nonzero(de)[0])
        d = (s[e] != s2[e])
        idx = torch.nonzero(d)[:6].tolist()
        print("  first env", e, "diff idx", idx, "vals", [(float(s[e][tuple(i)]), float(s2[e][tuple(i)])) for i in idx[:3]], "mask", m[e].tolist(), end="")
    print()
