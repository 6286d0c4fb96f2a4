"""One frozen state (after W steps), the sub-step kernel timed on it with the mixed waves on / off (TE_DENSE_MIN)."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, torch, numpy as np
sys.path.insert(0, %r)
from dronechase_amd import config as K, default_config
from dronechase_amd.batched_env import BatchedEnv
N, W, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
env = BatchedEnv(default_config("stage03", n_envs=N, seed=0), "cuda:0")
env.reset()
if mode == "make":
    a = torch.empty((N, 4), device="cuda:0")
    for t in range(W):
        env.random_actions(12345, t, out=a); env.step(a, terminal=False)
    torch.save(env.get_state().cpu(), "/tmp/frozen.pt")
    sys.exit(0)
w = torch.load("/tmp/frozen.pt").cuda()
a = env.random_actions(12345, W).clone()
D = env.D
armed = (w[: N * D * K.DRONE_WORDS].view(N, D, K.DRONE_WORDS)[:, :, K.D["ARMED"]] != 0)
k1 = []
for r in range(40):
    env.set_state(w); torch.cuda.synchronize()
    env.profile_begin(1); env.step(a, terminal=False); k = env.profile_end(); k1.append(k[0] * 1e3)
k1 = sorted(k1)
print(f"{mode:8s} armed/env {armed.float().sum(1).mean().item():.2f}  sub-step kernel median {k1[len(k1)//2]:.1f} us  min {k1[0]:.1f}")
'''
N, W = (sys.argv + ["65536", "1000"])[1:3]
subprocess.run([sys.executable, "-c", CHILD % root, N, W, "make"], check=True)
for name, extra in (("dense", {"TE_DENSE_MIN": "1"}), ("mixed60", {"TE_DENSE_MIN": "60"}), ("mixed32", {"TE_DENSE_MIN": "32"})):
    r = subprocess.run([sys.executable, "-c", CHILD % root, N, W, name], env={**os.environ, **extra}, capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-800:])
