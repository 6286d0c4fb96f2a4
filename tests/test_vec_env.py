"""Host logic of the SB3 VecEnv mirror (dronechase_amd/vec_env.py) against a stub backend on CPU, and the
real thing on the GPU."""
import numpy as np
import pytest

from dronechase_amd import config as K


class StubBackend:
    """Stands in for BatchedEnv: torch CPU tensors, scripted outputs."""

    def __init__(self, n):
        import torch
        self.device = torch.device("cpu")
        self.N = n
        self.lidar = torch.ones((n, 3, 13, 26)); self.inertial = torch.zeros((n, 15)); self.last_action = torch.zeros((n, 4))
        self.t_lidar = torch.full((n, 3, 13, 26), 0.5); self.t_inertial = torch.full((n, 15), 0.25); self.t_last_action = torch.zeros((n, 4))
        self.reward = torch.zeros(n); self.done = torch.zeros(n, dtype=torch.uint8); self.info = torch.zeros((n, 4), dtype=torch.int32)
        self.calls = []

    def reset(self, mask=None):
        self.calls.append("reset")
        return self.lidar, self.inertial, self.last_action

    def step(self, actions, terminal=True):
        import torch
        assert actions.shape == (self.N, 4) and actions.dtype == torch.float32
        self.calls.append("step")
        self.last_action = actions.clone()
        self.reward = actions[:, 3].clone()
        self.done = (actions[:, 0] > 0.5).to(torch.uint8)
        self.info = torch.arange(self.N * 4, dtype=torch.int32).reshape(self.N, 4)
        return self.lidar, self.inertial, self.last_action, self.reward, self.done, self.info

    def close(self):
        self.calls.append("close")


def test_vecenv_surface_with_stub():
    from dronechase_amd.vec_env import INFO_KEYS, ThreatEngageVecEnv
    n = 6
    v = ThreatEngageVecEnv("stage03", num_envs=n, backend=StubBackend(n))
    assert v.num_envs == n and v.render_mode is None and len(v.reset_infos) == n
    assert v.action_space.shape == (4,) and list(v.action_space.low) == [-1, -1, -1, 0]
    assert v.observation_space["lidar"].shape == (3, 13, 26) and v.observation_space["inertial_data"].shape == (15,)
    obs = v.reset()
    assert set(obs) == {"lidar", "inertial_data", "last_action"} and obs["lidar"].shape == (n, 3, 13, 26)
    assert isinstance(obs["lidar"], np.ndarray)
    a = np.zeros((n, 4), np.float32); a[2, 0] = 1.0; a[:, 3] = np.arange(n) / 10
    obs, rew, dones, infos = v.step(a)
    assert rew.dtype == np.float32 and dones.dtype == bool and list(dones) == [False, False, True, False, False, False]
    np.testing.assert_allclose(rew, np.arange(n) / 10, atol=1e-7)
    assert isinstance(infos, list) and len(infos) == n
    assert all(set(INFO_KEYS) <= set(i) and i["TimeLimit.truncated"] is False for i in infos)
    assert infos[1]["agent_kills"] == 4 and infos[1]["current_wave"] == 7
    assert "terminal_observation" in infos[2] and "terminal_observation" not in infos[0]
    t = infos[2]["terminal_observation"]
    assert t["lidar"].shape == (3, 13, 26) and float(t["lidar"][0, 0, 0]) == 0.5 and t["inertial_data"].shape == (15,)
    with pytest.raises(RuntimeError):
        v.step_wait()
    assert v.env_is_wrapped(object) == [False] * n and v.get_attr("render_mode") == [None] * n
    assert v.get_attr("num_envs", indices=[0, 1]) == [n, n] and v.seed(3) == [None] * n
    with pytest.raises(AttributeError):
        v.env_method("get_keymap")
    lazy = ThreatEngageVecEnv("stage03", num_envs=n, backend=StubBackend(n), infos="lazy")
    lazy.reset()
    _, _, _, li = lazy.step(a)
    assert len(li) == n and li[2]["terminal_observation"]["last_action"].shape == (4,) and li[0]["deads"] == 2
    v.close()
    with pytest.raises(ValueError):
        ThreatEngageVecEnv("stage03", num_envs=2, GUI=True, backend=StubBackend(2))


def test_make_config_maps_reference_kwargs():
    from dronechase_amd.vec_env import make_config
    c = make_config("stage03", 8, dome_radius=30, rl_frequency=30)
    assert c.dome_radius == 30 and c.lidar_radius == 60 and c.substeps == 8 and c.max_step == 300 and c.cooldown_steps == 60
    c = make_config("stage01", 8, rl_frequency=30)
    assert c.max_step == 600 and c.lidar_radius == 20  # 20 * rl_frequency; lidar radius hard-coded in level2
    assert make_config("stage02", 4, n_invaders=8).n_invaders == 8


def test_pipeline_factory_whitelists_kwargs():
    from dronechase_amd.envs import Exp03vFinalEnvironment
    from dronechase_amd.pipeline import ReinforcementLearningPipeline as RLP
    v = RLP.create_vectorized_environment(Exp03vFinalEnvironment, {"dome_radius": 25, "foo": 1}, n_envs=5, monitor=False,
                                          backend=StubBackend(5))
    assert v.num_envs == 5 and v.cfg.dome_radius == 25 and v.task == "exp03"
    with pytest.raises(ValueError):
        RLP.create_vectorized_environment(dict, {}, n_envs=2, backend=StubBackend(2))


def test_torch_output_reads_nothing_back_until_an_info_is_indexed():
    """output="torch": observations, rewards and dones stay tensors and the infos are a view that fetches done / info on first access
    (the on-device rollout never pays for them); infos="dicts" and infos="lazy" agree entry for entry."""
    import torch
    from dronechase_amd.vec_env import LazyInfos, ThreatEngageVecEnv
    n = 5
    a = torch.zeros((n, 4)); a[3, 0] = 1.0
    v = ThreatEngageVecEnv("stage03", num_envs=n, backend=StubBackend(n), output="torch", infos="lazy")
    v.reset()
    obs, rew, dones, infos = v.step(a)
    assert isinstance(obs["lidar"], torch.Tensor) and isinstance(rew, torch.Tensor) and dones.dtype == torch.bool
    assert isinstance(infos, LazyInfos) and callable(infos._info)          # not fetched yet
    assert len(infos) == n and not callable(infos._info)                    # ... now it is
    assert infos[3]["terminal_observation"]["lidar"].shape == (3, 13, 26) and "terminal_observation" not in infos[0]
    d = ThreatEngageVecEnv("stage03", num_envs=n, backend=StubBackend(n), output="torch", infos="dicts")
    d.reset()
    _, _, _, di = d.step(a)
    assert isinstance(di, list) and [set(x) for x in di] == [set(infos[i]) for i in range(n)]
    assert all(di[i][k] == infos[i][k] for i in range(n) for k in ("agent_kills", "allies_kills", "deads", "current_wave", "TimeLimit.truncated"))
    assert all(type(di[i]["deads"]) is int for i in range(n))


def test_lazy_infos_hold_the_values_of_their_own_step():
    """The backend's done / info tensors are persistent buffers that every step overwrites IN PLACE: infos indexed only after the next step
    has been issued (a logger that drains later) must still show the step that produced them."""
    import torch
    from dronechase_amd.vec_env import ThreatEngageVecEnv

    class InPlace(StubBackend):
        def step(self, actions, terminal=True):
            self.calls.append("step")
            self.last_action.copy_(actions); self.reward.copy_(actions[:, 3])
            self.done.copy_((actions[:, 0] > 0.5).to(torch.uint8))
            self.info.copy_((actions[:, 1:2] * 100).to(torch.int32).expand(-1, 4))
            return self.lidar, self.inertial, self.last_action, self.reward, self.done, self.info

    n = 4
    v = ThreatEngageVecEnv("stage03", num_envs=n, backend=InPlace(n), output="torch", infos="lazy")
    v.reset()
    a = torch.zeros((n, 4)); a[3, 0] = 1.0; a[:, 1] = 0.07
    b = torch.zeros((n, 4)); b[1, 0] = 1.0; b[:, 1] = 0.09
    _, _, _, first = v.step(a)
    _, _, _, second = v.step(b)
    assert first[0]["agent_kills"] == 7 and "terminal_observation" in first[3] and "terminal_observation" not in first[1]
    assert second[0]["agent_kills"] == 9 and "terminal_observation" in second[1] and "terminal_observation" not in second[3]


@pytest.mark.gpu
def test_lazy_infos_on_gpu_survive_the_next_step():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    from dronechase_amd.vec_env import ThreatEngageVecEnv
    n = 256
    v = ThreatEngageVecEnv("stage03", num_envs=n, seed=4, max_step=6, output="torch", infos="lazy")
    w = ThreatEngageVecEnv("stage03", num_envs=n, seed=4, max_step=6, output="torch", infos="dicts")
    v.reset(); w.reset()
    held, eager = [], []
    for s in range(16):
        a = v.backend.random_actions(9, s)
        held.append(v.step(a)[3]); eager.append(w.step(a)[3])
    saw_done = late_errors = 0
    for k, (lazy, now) in enumerate(zip(held, eager)):          # the lazy infos are only read here, 0..15 steps late
        last = k == len(held) - 1
        for e in range(n):
            ended = "terminal_observation" in now[e]
            assert lazy.has_terminal_observation(e) == ended
            saw_done += ended
            if ended and not last:
                # the counters of step k survive; its terminal ROWS do not (views of the backend's buffers): a late read must fail loudly
                # instead of handing out another step's rows (round-3 review)
                with pytest.raises(RuntimeError, match="LAST step"):
                    lazy[e]
                late_errors += 1
                continue
            got = lazy[e]
            assert {k2: got[k2] for k2 in ("agent_kills", "allies_kills", "deads", "current_wave")} == {k2: now[e][k2] for k2 in ("agent_kills", "allies_kills", "deads", "current_wave")}
            if ended:   # the last step's rows are still there, and they are the eager env's rows
                for key in ("lidar", "inertial_data", "last_action"):
                    assert torch.equal(got["terminal_observation"][key], now[e]["terminal_observation"][key])
    assert saw_done >= n and late_errors > 0
    v.close(); w.close()


@pytest.mark.gpu
def test_vecenv_on_gpu_matches_batched_env():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    from dronechase_amd import default_config
    from dronechase_amd.batched_env import BatchedEnv
    from dronechase_amd.envs import Exp03vFinalEnvironment, PyflytL2EnviromentModifiedV2
    from dronechase_amd.vec_env import ThreatEngageVecEnv

    n = 512
    v = ThreatEngageVecEnv("stage03", num_envs=n, seed=4, max_step=40)  # short episodes: auto-reset at step 41
    ref = BatchedEnv(default_config("stage03", n_envs=n, seed=4, max_step=40), "cuda:0")
    obs = v.reset(); ref.reset()
    assert (obs["lidar"] == 1).all()
    n_done = 0
    for s in range(150):
        a = ref.random_actions(9, s)
        obs, rew, dones, infos = v.step(a.cpu().numpy())
        l, i, la, r, d, info = ref.step(a)
        np.testing.assert_array_equal(obs["lidar"], l.cpu().numpy())
        np.testing.assert_array_equal(rew, r.cpu().numpy())
        np.testing.assert_array_equal(dones, d.cpu().numpy().astype(bool))
        for e in np.flatnonzero(dones):
            np.testing.assert_array_equal(infos[e]["terminal_observation"]["inertial_data"], ref.t_inertial[e].cpu().numpy())
            n_done += 1
        assert infos[0]["current_wave"] == int(info[0, 3])
    assert n_done > 0
    v.close(); ref.close()
    # single-env classes with the reference's names and (obs, info) / 5-tuple signatures
    env = Exp03vFinalEnvironment(dome_radius=20, rl_frequency=15, GUI=False)
    obs, info = env.reset(seed=0)
    assert obs["lidar"].shape == (3, 13, 26) and info == {}
    obs, reward, terminated, truncated, info = env.step(np.array([0, 0, 1, 0.5], np.float32))
    assert isinstance(reward, float) and truncated is False and set(info) == {"agent_kills", "allies_kills", "deads", "current_wave"}
    env.close()
    env = PyflytL2EnviromentModifiedV2()
    env.reset()
    total = 0
    for _ in range(305):
        obs, reward, terminated, truncated, info = env.step(np.array([0, 0, 0, 0], np.float32))
        total += 1
        if terminated:
            break
    assert terminated and total == 301  # step_calls > 300
    env.close()
