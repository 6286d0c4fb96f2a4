"""Per-step duration of the two te_step kernels along a bench rollout, with the armed-slot census that drives it."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dronechase_amd import default_config, config as K
from dronechase_amd.batched_env import BatchedEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 230
cfg = default_config("stage03", n_envs=n, motor_noise=1)
D = cfg.n_drones
env = BatchedEnv(cfg, torch.device("cuda", 0))
a = torch.empty((n, 4), device="cuda")
env.reset()
for i in range(steps):
    env.random_actions(1234, i, out=a)
    env.profile_begin(1)
    env.step(a, terminal=True)
    k1, k2, _ = env.profile_end()
    if i % 10 == 0 or i == steps - 1:
        w = env.get_state()
        armed = w[: n * D * K.DRONE_WORDS].view(n, D, K.DRONE_WORDS)[:, :, K.D["ARMED"]] != 0
        frac = armed.float().mean(0).tolist()
        live_waves = int(armed.view(n // 64, 64, D).any(1).sum().item())
        print(f"step {i:3d}: K1 {k1*1e3:6.1f} us K2 {k2*1e3:6.1f} us  active waves {live_waves}/{n // 64 * D}  armed/slot "
              + " ".join(f"{x:.2f}" for x in frac), flush=True)
env.close()
