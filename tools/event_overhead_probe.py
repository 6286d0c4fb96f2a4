import sys; sys.path.insert(0, "/root/repo")
import torch, numpy as np
from dronechase_amd import default_config
from dronechase_amd.batched_env import BatchedEnv
dev = torch.device("cuda", 0)
small = BatchedEnv(default_config("stage01", n_envs=64), dev)
a = torch.empty((64, 4), device=dev)
big = torch.empty((1 << 26,), device=dev)
for label, work in (("1-workgroup kernel", lambda: small.random_actions(1, 0, out=a)), ("nothing", lambda: None)):
    for busy in (False, True):
        ts = []
        for _ in range(200):
            if busy: big.fill_(1.0)            # the queue is not empty when the first marker arrives (as in a rollout)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); work(); e1.record()
            ts.append((e0, e1))
        torch.cuda.synchronize()
        el = np.array([x.elapsed_time(y) for x, y in ts]) * 1e3
        print(f"{label:20s} busy queue {busy}: median {np.median(el):.2f} us  p10 {np.percentile(el, 10):.2f}  p90 {np.percentile(el, 90):.2f}")
