"""Long rollout at full size: finite observations / rewards, bookkeeping identities, episode statistics."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dronechase_amd import default_config, config as K
from dronechase_amd.batched_env import BatchedEnv
task = sys.argv[1] if len(sys.argv) > 1 else "stage03"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
N = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
cfg = default_config(task, n_envs=N)
g = BatchedEnv(cfg, "cuda:0"); g.reset()
a = torch.empty((N, 4), device="cuda:0")
stacked = bool(cfg.stacked_obs)
dones = 0; rsum = 0.0; t0 = time.perf_counter()
for t in range(steps):
    g.random_actions(11, t, out=a)
    if int(cfg.ally_policy) == K.ALLY_EXTERNAL:  # exp05: the ally gets random commands too
        g.observe_ally(); g.set_ally_actions(g.random_actions(12, t).clone())
    out = g.step_stacked(a) if stacked else g.step(a)
    reward, done, info = out[-3], out[-2], out[-1]
    if t % 500 == 0 or t == steps - 1:
        obs = out[0]
        assert torch.isfinite(obs).all() and torch.isfinite(out[-5]).all() and torch.isfinite(reward).all(), t
        assert float(obs.min()) >= 0.0 and float(obs.max()) <= 1.0 and float(out[-5].abs().max()) <= 1.0 + 1e-6
        w = g.get_state()
        D = cfg.n_drones
        dr = w[: N * D * K.DRONE_WORDS].view(N, D, K.DRONE_WORDS)
        pos = dr[:, :, 0:3].view(torch.float32)
        assert torch.isfinite(pos).all() and float(pos.abs().max()) < 1e3, float(pos.abs().max())
        q = dr[:, :, 3:7].view(torch.float32)
        assert float((q.norm(dim=-1) - 1).abs().max()) < 1e-3
        armed = (dr[:, :, K.D["ARMED"]] != 0).float().sum(1).mean().item()
        print(f"step {t}: armed/env {armed:.2f}  wave {info[:, 3].float().mean().item():.2f}  dones so far {dones}  mean reward {rsum / max(t, 1):.2f}  "
              f"{N * (t + 1) / (time.perf_counter() - t0) / 1e6:.0f} M env-steps/s", flush=True)
    dones += int(done.sum()); rsum += float(reward.mean())
print("soak ok")
