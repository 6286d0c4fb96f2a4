"""Evaluation_Task rules on the MI355X (cfg.evaluation): te_step + te_wingman_info through the C ABI
against the oracle on identical seeded inputs, then the reference-shaped EvaluationEnvironment.  Tolerances as
tests/test_gpu_parity.py (STATE_TOL 1e-4 after one env.step from an identical state, OBS_TOL 1e-5, integers exact outside
the envs the oracle flags as ambiguous)."""
import numpy as np
import pytest

from tests.test_gpu_parity import MARGIN, OBS_TOL, STATE_TOL, _compare_states

pytestmark = pytest.mark.gpu


def _gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box (no CPU fallback exists)")
    return torch


@pytest.mark.parametrize("n_pursuers,limited", [(1, 0), (2, 1)])
def test_evaluation_single_step_parity_and_wingman_rows(n_pursuers, limited):
    torch = _gpu()
    from dronechase_amd import _lib, default_config
    from dronechase_amd.batched_env import BatchedEnv
    from oracle import te_oracle as O

    N = 2048
    rounds = int(_lib.load().te_calculate_rounds(n_pursuers, 20))
    cfg = default_config("evaluation", n_envs=N, motor_noise=1, seed=13, n_pursuers=n_pursuers, n_rounds=rounds, n_invaders=rounds,
                         max_step=300 if limited else 0)
    D = cfg.n_drones
    orc, gpu = O.OracleEnv(cfg, "f32", threads=8), BatchedEnv(cfg, "cuda:0")
    orc.reset(); gpu.reset()
    np.testing.assert_array_equal(gpu.wingman_info().cpu().numpy(), orc.wingman_info())
    zeros = np.zeros((N, 4), np.float32)
    noise = None
    step = n_ambiguous = kills = 0
    for chk in range(8):
        for _ in range(45):
            orc.step(zeros); step += 1
        gpu.set_state(torch.from_numpy(orc.get_state().view(np.int32)).cuda())
        noise = orc.random_actions(3, step)     # the action must be ignored: feed both sides something non-trivial
        ol, oi, oa, orew, odone, oinfo = (x.copy() for x in orc.step(noise))
        ok = orc.margins() > MARGIN
        gl, gi, ga, grew, gdone, ginfo = (x.cpu().numpy() for x in gpu.step(torch.from_numpy(noise).cuda()))
        diff, imis = _compare_states(orc.get_state(), gpu.get_state().cpu().numpy().view(np.uint32), N, D)
        n_ambiguous += int((~ok).sum())
        assert not (imis & ok).any() and not ((odone != gdone) & ok).any() and not ((oinfo != ginfo).any(1) & ok).any()
        good = ok & ~imis
        assert diff[good].max() < STATE_TOL, diff[good].max()
        assert (grew == 0).all() and (orew == 0).all()
        np.testing.assert_allclose(gi[good], oi[good], atol=OBS_TOL)
        rg, ro = gpu.wingman_info().cpu().numpy(), orc.wingman_info()
        np.testing.assert_array_equal(rg[good], ro[good])
        kills += int(ro[..., 0].sum())
        step += 1
    assert n_ambiguous <= 8 * N // 50 and kills > N // 4     # the behaviour tree does shoot invaders down
    gpu.close(); orc.close()


def test_evaluation_environment_surface_and_api_errors():
    """apps/threatengage_runner/stage03/experiments/01/evaluation_exp01_1bt_app_ready.py:60-96: the evaluation loop."""
    torch = _gpu()
    from dronechase_amd import _lib, default_config
    from dronechase_amd.batched_env import BatchedEnv
    from dronechase_amd.envs import EvaluationEnvironment

    env = EvaluationEnvironment({"SHOW_NAME": False, "drivers": [{"type": "bt", "name": "bt_1"}, {"type": "bt", "name": "bt_2"}]},
                                GUI=False, rl_frequency=15)
    assert env.cfg.n_pursuers == 2 and env.cfg.n_rounds == 9 and env.cfg.max_step == 0 and env.cfg.evaluation == 1
    observation, _ = env.reset(0)
    assert observation["lidar"].shape == (3, 13, 26)
    latest, terminated, steps = {}, False, 0
    while not terminated and steps < 3000:
        observation, reward, terminated, truncated, info = env.step(np.zeros(1))
        assert reward == 0.0 and truncated is False and set(info) <= {"bt_1", "bt_2"}
        for name, row in info.items():
            assert set(row) == {"lw_kills", "lw_alive", "lw_munitions", "current_wave", "step"} and row["lw_alive"] is True
            if name not in latest or row["step"] > latest[name]["step"]:      # update_data of the reference app
                latest[name] = row
        steps += 1
    # (an episode that ends with the last wingman's death has no row for its final step: the reference lists armed pursuers only)
    assert terminated and latest and max(r["step"] for r in latest.values()) >= steps - 1
    assert sum(r["lw_kills"] for r in latest.values()) >= 1
    env.close()
    plain = BatchedEnv(default_config("exp03", n_envs=64), "cuda:0")
    with pytest.raises(_lib.TEError, match="cfg.evaluation"):
        plain.wingman_info()
    plain.close()
    with pytest.raises(_lib.TEError, match="Evaluation_Task"):
        BatchedEnv(default_config("evaluation", n_envs=64, stacked_obs=1), "cuda:0")


def test_caller_driven_wingmen_parity_and_nn_driver_surface():
    """cfg.evaluation's driver mask (the reference's "nn" drivers, evaluation_task.py:257-268,655-661): te_observe_wingman /
    te_set_wingman_actions for pursuers 0 and 1 against the oracle, then EvaluationEnvironment with a model object."""
    torch = _gpu()
    from dronechase_amd import _lib, default_config
    from dronechase_amd.batched_env import BatchedEnv
    from dronechase_amd.envs import EvaluationEnvironment
    from oracle import te_oracle as O

    N, P = 4096, 2
    rounds = int(_lib.load().te_calculate_rounds(P, 20))
    cfg = default_config("evaluation", n_envs=N, motor_noise=1, seed=29, n_pursuers=P, n_rounds=rounds, n_invaders=rounds,
                         evaluation=1 | (0b11 << 8))
    D = cfg.n_drones
    orc, gpu = O.OracleEnv(cfg, "f32", threads=8), BatchedEnv(cfg, "cuda:0")
    orc.reset(); gpu.reset()
    zeros = np.zeros((N, 4), np.float32)
    step = n_ambiguous = 0
    for chk in range(6):
        for _ in range(25):
            for p in range(P):
                orc.set_wingman_actions(p, orc.random_actions(50 + p, step))
            orc.step(zeros); step += 1
        gpu.set_state(torch.from_numpy(orc.get_state().view(np.int32)).cuda())
        for p in range(P):
            ol, oi, oa, oact = orc.observe_wingman(p)
            gl, gi, ga, gact = (x.cpu().numpy() for x in gpu.observe_wingman(p))
            np.testing.assert_array_equal(gact, oact); np.testing.assert_array_equal(ga, oa)
            np.testing.assert_allclose(gi, oi, atol=OBS_TOL)
            assert (np.abs(gl - ol).reshape(N, -1).max(1) > OBS_TOL).sum() <= max(2, N // 500)
            act = orc.random_actions(50 + p, step)
            orc.set_wingman_actions(p, act); gpu.set_wingman_actions(p, torch.from_numpy(act).cuda())
        orc.step(zeros); gpu.step(torch.from_numpy(zeros).cuda()); step += 1
        ok = orc.margins() > MARGIN
        diff, imis = _compare_states(orc.get_state(), gpu.get_state().cpu().numpy().view(np.uint32), N, D)
        n_ambiguous += int((~ok).sum())
        assert not (imis & ok).any()
        assert diff[ok & ~imis].max() < STATE_TOL
        np.testing.assert_array_equal(gpu.wingman_info().cpu().numpy()[ok & ~imis], orc.wingman_info()[ok & ~imis])
    assert n_ambiguous <= 6 * N // 50
    gpu.close(); orc.close()
    scripted = BatchedEnv(default_config("evaluation", n_envs=64), "cuda:0")
    with pytest.raises(_lib.TEError, match="not driven by the caller"):
        scripted.observe_wingman(0)
    scripted.close()
    with pytest.raises(_lib.TEError, match="does not exist"):
        BatchedEnv(default_config("evaluation", n_envs=64, evaluation=1 | (0b10 << 8)), "cuda:0")

    class Model:                      # what PPO.load would return in the reference
        calls = 0

        def predict(self, observation, deterministic=True):
            assert deterministic and observation["lidar"].shape == (3, 13, 26) and observation["inertial_data"].shape == (15,)
            Model.calls += 1
            return np.array([1.0, 0.0, 0.0, 0.5], np.float32), None

    env = EvaluationEnvironment({"drivers": [{"type": "nn", "name": "nn_1", "model": Model()}, {"type": "bt", "name": "bt_1"}]})
    assert env.cfg.evaluation == 1 | (0b01 << 8)
    env.reset(0)
    for _ in range(5):
        obs, reward, terminated, truncated, info = env.step(np.zeros(1))
    assert Model.calls == 5 and set(info) <= {"nn_1", "bt_1"} and reward == 0.0
    from dronechase_amd import config as K
    sp = env._b.get_state().view(torch.float32)[: env._b.D * K.DRONE_WORDS].view(env._b.D, K.DRONE_WORDS)[0, K.D["SETPOINT"]:K.D["SETPOINT"] + 4]
    np.testing.assert_allclose(sp.cpu().numpy(), [0.5, 0.0, 0.0, 0.0], atol=1e-6)     # the model's command, not the behaviour tree's
    env.close()
    with pytest.raises(ValueError, match="model"):
        EvaluationEnvironment({"drivers": [{"type": "nn", "path": "model.zip", "name": "nn_1"}]})
