"""On-device PPO (dronechase_amd/ppo.py; BASELINE.json config 5): the pieces that need no GPU, and a short training run
on the MI355X."""
import numpy as np
import pytest


def test_policy_topology_matches_the_reference_extractor():
    """ppo_policies.py:234-341: conv(k4,s4,32) -> conv(k2,s2,64) on [3,13,26] flattens to 64*1*3 = 192; three 128-wide
    layers for inertial_data and last_action; 448 -> 256; SB3 default 2 x 64 tanh heads."""
    import torch
    from dronechase_amd.ppo import LidarInertialActionPolicy
    p = LidarInertialActionPolicy()
    obs = {"lidar": torch.rand(5, 3, 13, 26), "inertial_data": torch.rand(5, 15) * 2 - 1, "last_action": torch.rand(5, 4)}
    assert p.lidar(obs["lidar"]).shape == (5, 192) and p.final[0].in_features == 192 + 128 + 128 and p.final[0].out_features == 256
    mu, v = p(obs)
    assert mu.shape == (5, 4) and v.shape == (5,)
    d, _ = p.dist(obs)
    assert d.log_prob(mu).sum(-1).shape == (5,) and torch.allclose(d.stddev, torch.ones(5, 4))


def test_gae_against_a_naive_loop():
    import torch
    from dronechase_amd.ppo import RolloutBuffer
    T, N, g, lam = 7, 5, 0.99, 0.95
    rng = np.random.default_rng(0)
    b = RolloutBuffer(T, N, {"lidar": (3, 13, 26), "inertial_data": (15,), "last_action": (4,)}, "cpu")
    r, v, d = rng.normal(size=(T, N)), rng.normal(size=(T, N)), (rng.random((T, N)) < 0.2).astype(np.float64)
    last = rng.normal(size=N)
    b.rewards.copy_(torch.tensor(r)); b.values.copy_(torch.tensor(v)); b.dones.copy_(torch.tensor(d))
    b.finish(torch.tensor(last, dtype=torch.float32), g, lam)
    adv = np.zeros((T, N))
    for n in range(N):
        for t in range(T):
            acc, disc = 0.0, 1.0
            for k in range(t, T):
                nv = last[n] if k == T - 1 else v[k + 1, n]
                delta = r[k, n] + g * nv * (1 - d[k, n]) - v[k, n]
                acc += disc * delta
                if d[k, n]:
                    break
                disc *= g * lam
            adv[t, n] = acc
    np.testing.assert_allclose(b.adv.numpy(), adv, atol=1e-5)
    np.testing.assert_allclose(b.ret.numpy(), adv + v, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("use_graph", [True, False])
def test_short_training_run_on_device(use_graph):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    from dronechase_amd import default_config
    from dronechase_amd.batched_env import BatchedEnv
    from dronechase_amd.ppo import PPO, PPOConfig
    env = BatchedEnv(default_config("stage03", n_envs=512, max_step=40), "cuda:0")
    ppo = PPO(env, PPOConfig(n_steps=16, batch_size=2048, n_epochs=2, use_graph=use_graph), seed=1)
    before = [p.detach().clone() for p in ppo.policy.parameters()]
    logs = []
    ppo.learn(2 * 16 * 512, log=logs.append)
    assert len(logs) == 2 and ppo.num_timesteps == 2 * 16 * 512
    for entry in logs:
        assert all(np.isfinite(v) for v in entry.values()), entry
        assert 0.0 <= entry["clip_frac"] <= 1.0 and entry["entropy"] > 0
    assert logs[-1]["episodes_finished"] >= 0
    assert any(not torch.equal(a, b) for a, b in zip(before, ppo.policy.parameters()))
    # the rollout never left the device
    assert all(t.device.type == "cuda" for t in ppo.buf.obs.values()) and ppo.buf.adv.device.type == "cuda"
    # the buffer holds a real rollout: consecutive observations differ, actions were stored, the step counter of the
    # environments advanced by exactly the collected steps (the graph's warm-up and capture steps were rolled back)
    b = ppo.buf
    assert bool((b.obs["inertial_data"][0] != b.obs["inertial_data"][5]).any()) and bool((b.actions.abs().sum((1, 2)) > 0).all())
    assert bool(torch.isfinite(b.values).all()) and bool(torch.isfinite(b.logp).all())
    from dronechase_amd import config as K
    w = env.get_state().view(torch.int32)
    steps = w[512 * env.D * K.DRONE_WORDS:].view(512, K.ENV_WORDS)[:, K.E["STEP"]]
    assert int(steps.max()) <= 2 * 16 and int(steps.max()) >= 16
    env.close()
