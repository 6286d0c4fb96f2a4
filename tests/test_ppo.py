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


class _StubEnv:
    """What PPO needs of an environment to build its buffers (no stepping): used where there is no GPU."""

    def __init__(self, n):
        import torch
        self.device, self.N = torch.device("cpu"), n
        self.lidar = torch.ones((n, 3, 13, 26)); self.inertial = torch.zeros((n, 15)); self.cfg = None

    def reset(self):
        return None


def _ppo_rank(rank, world, port, out_dir):
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    from dronechase_amd.ppo import PPO, PPOConfig
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)

    def fill(ppo, seed):   # a synthetic rollout, different on every rank (each rank owns its own env shard)
        g = torch.Generator().manual_seed(seed)
        b = ppo.buf
        b.obs["lidar"].copy_(torch.rand(b.obs["lidar"].shape, generator=g)); b.obs["inertial_data"].copy_(torch.rand(b.obs["inertial_data"].shape, generator=g) * 2 - 1)
        b.obs["last_action"].copy_(torch.rand(b.obs["last_action"].shape, generator=g))
        b.actions.copy_(torch.randn(b.actions.shape, generator=g)); b.logp.copy_(-torch.rand(b.logp.shape, generator=g) * 4)
        b.values.copy_(torch.randn(b.values.shape, generator=g)); b.rewards.copy_(torch.randn(b.rewards.shape, generator=g))
        b.dones.copy_((torch.rand(b.dones.shape, generator=g) < 0.1).float())
        b.finish(torch.randn(b.rewards.shape[1], generator=g), 0.99, 0.95)

    cfg = PPOConfig(n_steps=4, batch_size=32, n_epochs=2, use_graph=False)
    # (1) what a rank would learn on its own data alone (no process group yet)
    alone = PPO(_StubEnv(16), cfg, seed=5)
    fill(alone, 100 + rank)
    torch.manual_seed(7); alone.update()
    # (2) the data-parallel run: rank-dependent initial weights on purpose, PPO must broadcast rank 0's
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ppo = PPO(_StubEnv(16), cfg, seed=5 + rank)
    assert ppo.distributed
    init = torch.cat([p.detach().flatten() for p in ppo.policy.parameters()]).clone()
    fill(ppo, 100 + rank)
    torch.manual_seed(7); stats = ppo.update()
    flat = torch.cat([p.detach().flatten() for p in ppo.policy.parameters()])
    bucket = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(bucket, flat)
    inits = [torch.empty_like(init) for _ in range(world)]
    dist.all_gather(inits, init)
    torch.save(dict(params=bucket, inits=inits, alone=torch.cat([p.detach().flatten() for p in alone.policy.parameters()]), stats=stats,
                    views=all(p.grad.data_ptr() >= ppo._flat_grad.data_ptr() for p in ppo.policy.parameters())), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_data_parallel_update_over_gloo(tmp_path):
    """BASELINE config 5's only collective, on the CPU with two gloo ranks: PPO.update() on different rollouts per rank must leave
    IDENTICAL parameters on both ranks (initial broadcast + one bucketed gradient all-reduce per minibatch), different from the
    initial ones and different from what either rank learns alone."""
    import socket
    import torch
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.start_processes(_ppo_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt", weights_only=False) for r in (0, 1))
    assert torch.equal(r0["inits"][0], r0["inits"][1])                      # rank 0's weights everywhere before the first step
    assert torch.equal(r0["params"][0], r0["params"][1]) and torch.equal(r0["params"][0], r1["params"][1])   # ... and after the update
    assert not torch.equal(r0["params"][0], r0["inits"][0])
    assert not torch.allclose(r0["params"][0], r0["alone"]) and not torch.allclose(r1["params"][1], r1["alone"])
    assert r0["views"] and all(np.isfinite(v) for v in r0["stats"].values())


@pytest.mark.gpu
def test_training_run_at_16k_envs():
    """Config 5 at a size that means something on one GPU: 16 384 stage03 envs, 32-step rollouts captured in a HIP graph (1.1 GB of
    observations resident in HBM), two updates of 16 minibatches."""
    import time
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    from dronechase_amd import default_config
    from dronechase_amd.batched_env import BatchedEnv
    from dronechase_amd.ppo import PPO, PPOConfig
    N = 16384
    env = BatchedEnv(default_config("stage03", n_envs=N), "cuda:0")
    ppo = PPO(env, PPOConfig(n_steps=32, batch_size=32768, n_epochs=1, use_graph=True), seed=3)
    assert ppo.buf.bytes() > 2 ** 30
    logs = []
    ppo.collect(); ppo.update()        # captures the graph; MIOpen picks its convolution algorithms for both batch shapes (seconds, once)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ppo.learn(ppo.num_timesteps + 2 * 32 * N, log=logs.append)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    assert len(logs) == 2 and all(np.isfinite(v) for e in logs for v in e.values())
    assert logs[-1]["episodes_finished"] >= 0 and 0.0 <= logs[-1]["clip_frac"] <= 1.0
    print(f"PPO at {N} envs: {2 * 32 * N / dt / 1e6:.2f} M env-steps/s collect + update")
    env.close()


@pytest.mark.gpu
def test_fast_learner_switch_trains_with_finite_losses():
    """PPOConfig.fast_learner (fused Adam + bf16 autocast of the policy's forward / backward, fp32 master weights and losses): two updates on
    real rollouts of 4 096 stage03 envs; every statistic finite, the parameters move, and they stay fp32."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    from dronechase_amd import default_config
    from dronechase_amd.batched_env import BatchedEnv
    from dronechase_amd.ppo import PPO, PPOConfig
    env = BatchedEnv(default_config("stage03", n_envs=4096), "cuda:0")
    ppo = PPO(env, PPOConfig(n_steps=16, batch_size=8192, n_epochs=2, use_graph=True, fast_learner=True), seed=5)
    before = torch.cat([p.detach().flatten() for p in ppo.policy.parameters()]).clone()
    logs = []
    ppo.learn(2 * 16 * 4096, log=logs.append)
    after = torch.cat([p.detach().flatten() for p in ppo.policy.parameters()])
    assert len(logs) == 2 and all(np.isfinite(v) for e in logs for v in e.values())
    assert torch.isfinite(after).all() and not torch.equal(before, after)
    assert all(p.dtype == torch.float32 for p in ppo.policy.parameters())
    assert 0.0 <= logs[-1]["clip_frac"] <= 1.0 and logs[-1]["entropy"] > 0
    env.close()
