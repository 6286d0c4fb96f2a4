"""Host-side cost of one bench step (ctypes + launches): time the bench loop on a tiny shard where the GPU work is negligible."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dronechase_amd import default_config
from dronechase_amd.batched_env import BatchedEnv
for n in (64, 4096):
    cfg = default_config("stage03", n_envs=n)
    env = BatchedEnv(cfg, torch.device("cuda", 0))
    a = torch.empty((n, 4), device="cuda")
    env.reset()
    for i in range(50):
        env.random_actions(1, i, out=a); env.step(a, terminal=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(2000):
        env.random_actions(1, i, out=a); env.step(a, terminal=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"n={n}: host enqueue {1e6*(t1-t0)/2000:.1f} us/step, with sync {1e6*(t2-t0)/2000:.1f} us/step")
    t0 = time.perf_counter()
    for i in range(2000):
        env.step(a, terminal=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"n={n}: step only enqueue {1e6*(t1-t0)/2000:.1f} us/step")
    env.close()
