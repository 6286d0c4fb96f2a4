"""exp05 on the MI355X: te_observe_ally / te_set_ally_actions / te_step through the C ABI
against the oracle on identical seeded inputs, then the reference-shaped surface (VecEnv.update_model, the single-env
class).  Tolerances as tests/test_gpu_parity.py: STATE_TOL 1e-4 after one env.step from an identical state, OBS_TOL 1e-5
on normalised observations, integers exact outside the envs the oracle flags as ambiguous (MARGIN 1e-4 m)."""
import numpy as np
import pytest

from tests.test_gpu_parity import MARGIN, OBS_TOL, STATE_TOL, _compare_states

pytestmark = pytest.mark.gpu


def _gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box (no CPU fallback exists)")
    return torch


@pytest.mark.parametrize("noise", [0, 1])
def test_exp05_single_step_parity_with_a_driven_ally(noise):
    torch = _gpu()
    from dronechase_amd import default_config
    from dronechase_amd.batched_env import BatchedEnv
    from oracle import te_oracle as O

    N = 4096 if noise else 2048     # 4 096: te_observe_ally's two-launch path (background waves + owner planes, then patches); below: one launch
    cfg = default_config("exp05", n_envs=N, motor_noise=noise, seed=31)
    D = cfg.n_drones
    orc, gpu = O.OracleEnv(cfg, "f32", threads=8), BatchedEnv(cfg, "cuda:0")
    orc.reset(); gpu.reset()
    for x, y in zip(gpu.observe_ally(), orc.observe_ally()):   # reset: empty sphere, zero last action, every ally alive
        np.testing.assert_allclose(x.cpu().numpy(), y, atol=OBS_TOL)
    step = n_ambiguous = dead_allies = lidar_flips = 0
    for chk in range(8):
        for _ in range(29):
            orc.set_ally_actions(orc.random_actions(41, step))
            orc.step(orc.random_actions(23, step)); step += 1
        gpu.set_state(torch.from_numpy(orc.get_state().view(np.int32)).cuda())
        # (1) the ally's observation of the identical state
        ol, oi, oa, oact = orc.observe_ally()
        gl, gi, ga, gact = (x.cpu().numpy() for x in gpu.observe_ally())
        np.testing.assert_array_equal(gact, oact); np.testing.assert_array_equal(ga, oa)
        np.testing.assert_allclose(gi, oi, atol=OBS_TOL)
        bad = np.abs(gl - ol).reshape(N, -1).max(1) > OBS_TOL      # a cell index may flip at a cell edge (as the agent's sphere)
        lidar_flips += int(bad.sum())
        assert bad.sum() <= max(2, N // 500)
        assert (gl[:, 0] < 1).any() and np.isclose(gl[:, 1], 0.6).any()   # the agent shows up as a wingman
        dead_allies += int((oact == 0).sum())
        # (2) drive + step
        ally, a = orc.random_actions(41, step), orc.random_actions(23, step); step += 1
        orc.set_ally_actions(ally); gpu.set_ally_actions(torch.from_numpy(ally).cuda())
        ol, oi, oa, orew, odone, oinfo = (x.copy() for x in orc.step(a))
        ok = orc.margins() > MARGIN
        gl, gi, ga, grew, gdone, ginfo = (x.cpu().numpy() for x in gpu.step(torch.from_numpy(a).cuda()))
        diff, imis = _compare_states(orc.get_state(), gpu.get_state().cpu().numpy().view(np.uint32), N, D)
        n_ambiguous += int((~ok).sum())
        assert not (imis & ok).any() and not ((odone != gdone) & ok).any() and not ((oinfo != ginfo).any(1) & ok).any()
        good = ok & ~imis
        assert diff[good].max() < STATE_TOL, diff[good].max()
        np.testing.assert_allclose(grew[good], orew[good], rtol=1e-5, atol=1e-3)
        np.testing.assert_allclose(gi[good], oi[good], atol=OBS_TOL)
        # (3) after the step: last action = what was sent where the ally was driven, zeros where the env auto-reset
        _, _, ga2, _ = (x.cpu().numpy() for x in gpu.observe_ally())
        _, _, oa2, _ = orc.observe_ally()
        np.testing.assert_array_equal(ga2[good], oa2[good])
        assert (ga2[good & (gdone != 0)] == 0).all()
    assert n_ambiguous <= 8 * N // 50 and dead_allies > 0  # some allies had died: the "not driven" branch was taken
    gpu.close(); orc.close()


def test_exp05_rollout_determinism_and_api_errors():
    torch = _gpu()
    from dronechase_amd import _lib, default_config
    from dronechase_amd.batched_env import BatchedEnv

    N = 65536
    cfg = default_config("exp05", n_envs=N, seed=3)
    outs = []
    for _ in range(2):
        g = BatchedEnv(cfg, "cuda:0")
        g.reset()
        for t in range(10):
            lidar, inertial, last_action, active = g.observe_ally()
            g.set_ally_actions(g.random_actions(19, t).clone())
            res = g.step(g.random_actions(7, t), terminal=False)
        outs.append([x.clone() for x in res] + [x.clone() for x in g.observe_ally()] + [g.get_state().clone()])
        g.close()
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    lidar, active = outs[0][6], outs[0][9]
    assert bool(((lidar >= 0) & (lidar <= 1)).all()) and int(active.sum()) > N // 2
    plain = BatchedEnv(default_config("exp03", n_envs=64), "cuda:0")
    with pytest.raises(_lib.TEError, match="TE_ALLY_EXTERNAL"):
        plain.observe_ally()
    with pytest.raises(_lib.TEError, match="TE_ALLY_EXTERNAL"):
        plain.set_ally_actions(torch.zeros((64, 4), device="cuda:0"))
    with pytest.raises(_lib.TEError, match="2 pursuers"):
        BatchedEnv(default_config("exp05", n_envs=64, n_pursuers=3), "cuda:0")
    plain.close()


def test_exp05_vecenv_driver_and_single_env_surface():
    """The reference's usage (apps/threatengage_runner/stage03/experiments/05/bo_exp05_vFinal_home_office_app.py:140-179):
    the ally flies a copy of a policy handed over with update_model / env_method."""
    torch = _gpu()
    from dronechase_amd.envs import Exp05vFinalEnvironment
    from dronechase_amd.pipeline import ReinforcementLearningPipeline
    from dronechase_amd.ppo import LidarInertialActionPolicy, PolicyDriver

    n = 128
    v = ReinforcementLearningPipeline.create_vectorized_environment(Exp05vFinalEnvironment, {"dome_radius": 20, "rl_frequency": 15},
                                                                    n_envs=n, monitor=False)
    v.reset()
    a = np.tile(np.array([[0.3, -0.2, 0.1, 0.5]], np.float32), (n, 1))
    with pytest.raises(AttributeError, match="update_model"):
        v.step(a)
    torch.manual_seed(0)
    policy = LidarInertialActionPolicy().to("cuda:0")
    v.env_method("update_model", PolicyDriver(policy))          # on-device driver: observations stay in HBM

    class Numpy:                                                 # an SB3-style model gets numpy batches
        calls = 0

        def predict(self, observation, deterministic=True):
            assert deterministic and isinstance(observation["lidar"], np.ndarray) and observation["lidar"].shape == (n, 3, 13, 26)
            assert observation["inertial_data"].shape == (n, 15) and observation["last_action"].shape == (n, 4)
            Numpy.calls += 1
            return np.tile(np.array([[1.0, 0.0, 0.0, 1.0]], np.float32), (n, 1)), None

    for t in range(3):
        obs, rew, dones, infos = v.step(a)
    from dronechase_amd import config as K
    st = v.backend.get_state().view(torch.float32)[: n * v.backend.D * K.DRONE_WORDS].view(n, v.backend.D, K.DRONE_WORDS)
    sp = st[:, 1, K.D["SETPOINT"]:K.D["SETPOINT"] + 4]
    assert bool((sp.abs().sum(1) > 0).any()) and bool((sp.norm(dim=1) <= 1.0 + 1e-5).all())   # the policy's command, |v| <= magnitude <= 1
    v.update_model(Numpy())
    v.step(a)
    assert Numpy.calls == 1
    st = v.backend.get_state().view(torch.float32)[: n * v.backend.D * K.DRONE_WORDS].view(n, v.backend.D, K.DRONE_WORDS)
    np.testing.assert_allclose(st[:, 1, K.D["SETPOINT"]:K.D["SETPOINT"] + 4].cpu().numpy(), np.tile([[1.0, 0, 0, 0]], (n, 1)), atol=1e-6)
    assert set(obs) == {"lidar", "inertial_data", "last_action"}
    v.close()
    e = Exp05vFinalEnvironment(dome_radius=20, rl_frequency=15)
    e.reset()
    with pytest.raises(AttributeError, match="update_model"):
        e.step(np.zeros(4, np.float32))

    class One:
        def predict(self, observation, deterministic=True):
            assert observation["lidar"].shape == (3, 13, 26) and observation["inertial_data"].shape == (15,)
            return np.array([0, 0, 1, 0.5], np.float32), None

    e.update_model(One())
    o, r, term, trunc, info = e.step(np.array([0, 0, 1, 0.5], np.float32))
    assert o["lidar"].shape == (3, 13, 26) and trunc is False and set(info) >= {"agent_kills", "allies_kills", "deads", "current_wave"}
    e.close()
