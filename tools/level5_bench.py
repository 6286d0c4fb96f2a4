"""level5 throughput (stacked observation, 24 KB of observation per env-step): wall clock over a short rollout."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dronechase_amd import default_config
from dronechase_amd.batched_env import BatchedEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
g = BatchedEnv(default_config("level5", n_envs=N), "cuda:0")
g.reset()
a = torch.empty((N, 4), device="cuda:0")
for t in range(20):
    g.random_actions(5, t, out=a); g.step_stacked(a)
torch.cuda.synchronize(); t0 = time.perf_counter()
for t in range(steps):
    g.random_actions(5, 20 + t, out=a); g.step_stacked(a)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"level5 {N} envs: {N * steps / dt / 1e6:.1f} M env-steps/s, {1e6 * dt / steps:.0f} us/step, obs stream {N * 24336 * steps / dt / 1e12:.2f} TB/s")
