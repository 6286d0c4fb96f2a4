"""The engagement / termination / wave / navigator scenarios of tests/test_oracle_tasks.py replayed through
the C ABI on the GPU and compared with the oracle output for output (rare branches that random rollouts
seldom reach: explosions, suicides, double shots, origin removal, floors, last wave, auto-reset)."""
import numpy as np
import pytest

from tests._blob import Blob

pytestmark = pytest.mark.gpu

SCENARIOS = {
    # name: (task, cfg overrides, arena kwargs, blob tweaks [(kind, args)], n_steps, action)
    "shoot_hit": ("exp03", dict(hit_prob=1.0), dict(), [], 2, [0, 0, 0, 0]),
    "shoot_miss": ("exp03", dict(hit_prob=0.0), dict(), [], 2, [0.3, 0.1, 0, 0.5]),
    "cooldown": ("exp03", dict(hit_prob=1.0), dict(), [("ei", "STEP", 100), ("di", 0, "LAST_FIRED", 90), ("di", 0, "MUNITION", 19)], 2, [0, 0, 0, 0]),
    "explosion": ("exp03", dict(hit_prob=0.0), dict(invaders=((0.1, 0, 3),)), [], 1, [0, 0, 0, 0]),
    "explosion_noreset": ("exp03", dict(hit_prob=0.0, auto_reset=0), dict(invaders=((0.1, 0, 3),)), [], 1, [0, 0, 0, 0]),
    "suicide": ("exp03", dict(hit_prob=1.0), dict(invaders=((0.1, 0, 3),)), [("di", 0, "MUNITION", 0)], 1, [0, 0, 0, 0]),
    "ally_suicide": ("exp03", dict(hit_prob=1.0), dict(ally=(5, 0, 3), invaders=((5.1, 0, 3), (0, 5, 3))), [("di", 1, "MUNITION", 0)], 2, [0, 0, 0, 0]),
    "ally_kill": ("exp03", dict(hit_prob=1.0), dict(ally=(5, 0, 3), invaders=((5.5, 0, 3),)), [], 2, [0, 0, 0, 0]),
    "double_shot": ("exp03", dict(hit_prob=1.0), dict(ally=(1.0, 0, 3), invaders=((0.5, 0, 3),)), [], 2, [0, 0, 0, 0]),
    "origin": ("exp03", dict(), dict(agent=(0, 3, 3), invaders=((0.05, 0, 0.05), (4, 0, 3))), [], 2, [0, 0, 0, 0]),
    "dome_agent": ("exp03", dict(), dict(agent=(0, 20.5, 3), invaders=((0, 5, 3),)), [], 1, [0, 0, 0, 0]),
    "dome_invader": ("exp03", dict(), dict(invaders=((0, 20.5, 3),)), [], 1, [0, 0, 0, 0]),
    "low": ("exp03", dict(), dict(agent=(0, 0, -5.5), invaders=((0, 3, 3),)), [], 2, [0, 0, 1, 1]),
    "floor": ("exp03", dict(), dict(agent=(0, 0, -6.2), invaders=((0, 3, 3),)), [], 1, [0, 0, 0, 0]),
    "zone": ("exp03", dict(), dict(agent=(5, 0, 0), ally=(5, 1.5, 0), invaders=((5, 3, 0),)), [], 3, [1, 0, 0, 1]),
    "step_limit": ("exp03", dict(), dict(invaders=((0, 5, 3),)), [("ei", "STEP", 299)], 3, [0, 0, 0, 0]),
    "last_wave": ("exp03", dict(hit_prob=1.0), dict(), [("ei", "ROUND", 9)], 1, [0, 0, 0, 0]),
    "chase": ("exp03", dict(), dict(ally=(0, 3, 3), invaders=((6, 0, 3),)), [], 6, [0.3, 1, 0, 0.7]),
    "ally_dead": ("exp03", dict(), dict(ally=(0, 3, 3), invaders=((6, 0, 3), (0, 6, 3))), [("di", 1, "ARMED", 0), ("snap",)], 4, [0, 1, 0, 0.7]),
    "exp02_hit": ("exp02", dict(hit_prob=1.0), dict(), [], 3, [0, 0, 0, 0]),
    "exp04_frozen": ("exp04", dict(), dict(ally=(0, 3, 3), invaders=((6, 0, 3),)), [], 4, [1, 0, 0, 1]),
    "cone_nav": ("exp03", dict(kamikaze_cone_check=1), dict(agent=(0, 5, 5), ally=(0, 10, 5), invaders=((0, 0, 10), (0, 3, 8))), [], 5, [0, 0, 0, 0]),
}


def _build(name):
    from oracle import te_oracle as O
    task, over, arena_kw, tweaks, n_steps, action = SCENARIOS[name]
    over = dict(over); over.setdefault("motor_noise", 0)
    cfg = O.default_config(task, n_envs=1, **over)
    orc = O.OracleEnv(cfg, "f32")
    orc.reset()
    b = Blob(orc.get_state(), 1, cfg.n_drones)
    P = cfg.n_pursuers
    agent = arena_kw.get("agent", (0, 0, 3)); ally = arena_kw.get("ally", (3, 3, 3)); inv = arena_kw.get("invaders", ((0.5, 0, 3),))
    b.place(0, 0, agent); b.hover_ready(0, 0, cfg)
    if P > 1:
        b.place(0, 1, ally); b.hover_ready(0, 1, cfg)
    for j in range(cfg.n_invaders):
        if j < len(inv):
            b.place(0, P + j, inv[j]); b.hover_ready(0, P + j, cfg)
        else:
            b.place(0, P + j, (50, 50, 50), armed=0)
    b.set_ei(0, "ROUND", max(1, len(inv)))
    for t in tweaks:
        if t[0] == "ei": b.set_ei(0, t[1], t[2])
        elif t[0] == "di": b.set_i(0, t[1], t[2], t[3])
    b.refresh_snapshot(0)
    return cfg, orc, b, n_steps, action


@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_scenario_matches_oracle(name):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    from dronechase_amd.batched_env import BatchedEnv
    from tests.test_gpu_parity import _compare_states

    cfg, orc, b, n_steps, action = _build(name)
    gpu = BatchedEnv(cfg, "cuda:0")
    orc.set_state(b.w)
    gpu.set_state(torch.from_numpy(b.w.view(np.int32)).cuda())
    a = np.asarray([action], np.float32)
    for s in range(n_steps):
        ol, oi, oa, orew, odone, oinfo = (x.copy() for x in orc.step(a))
        gl, gi, ga, grew, gdone, ginfo = (x.cpu().numpy() for x in gpu.step(torch.from_numpy(a).cuda()))
        assert orc.margins()[0] > 1e-4, "scenario must not sit on a threshold"
        np.testing.assert_array_equal(gdone, odone)
        np.testing.assert_array_equal(ginfo, oinfo)
        np.testing.assert_allclose(grew, orew, rtol=1e-5, atol=1e-3)
        np.testing.assert_allclose(gi, oi, atol=1e-5)
        np.testing.assert_allclose(gl, ol, atol=1e-5)
        np.testing.assert_allclose(ga, oa, atol=0)
        if odone[0] and cfg.auto_reset:
            np.testing.assert_allclose(gpu.t_lidar.cpu().numpy(), orc.t_lidar, atol=1e-5)
            np.testing.assert_allclose(gpu.t_inertial.cpu().numpy(), orc.t_inertial, atol=1e-5)
            np.testing.assert_allclose(gpu.t_last_action.cpu().numpy(), orc.t_last_action, atol=0)
        diff, imis = _compare_states(orc.get_state(), gpu.get_state().cpu().numpy().view(np.uint32), 1, cfg.n_drones)
        assert not imis.any(), f"{name}: integer state differs at step {s}"
        assert diff.max() < 1e-4, f"{name}: state diff {diff.max():.2e} at step {s}"
    gpu.close(); orc.close()


def test_stage_scenarios_match_oracle():
    """stage01 catch/respawn (with the pending wrench) and stage02 suicide-kill/respawn/explosion."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    from dronechase_amd.batched_env import BatchedEnv
    from oracle import te_oracle as O
    from tests.test_gpu_parity import _compare_states

    for task in ("stage01", "stage02"):
        for noise in (0, 1):
            cfg = O.default_config(task, n_envs=1, motor_noise=noise, seed=3)
            orc = O.OracleEnv(cfg, "f32"); orc.reset()
            b = Blob(orc.get_state(), 1, cfg.n_drones)
            P = cfg.n_pursuers
            if task == "stage01":
                b.place(0, 0, (0.2, 0, 1)); b.place(0, 1, (3, 3, 3)); b.place(0, 2, (0, 0, 1))
                b.set_f(0, 2, "SETPOINT", [0, 0, 0, 1])
            else:
                b.place(0, 0, (0, 0, 3)); b.set_i(0, 0, "MUNITION", 0); b.place(0, 1, (1, 1, 1))
                for j in range(cfg.n_invaders):
                    b.place(0, P + j, (0.5, 0, 3) if j == 0 else (4 * np.cos(1.2 * j), 4 * np.sin(1.2 * j), 2 + 0.5 * j))
            for d in range(cfg.n_drones):
                b.hover_ready(0, d, cfg)
            b.refresh_snapshot(0)
            gpu = BatchedEnv(cfg, "cuda:0")
            orc.set_state(b.w); gpu.set_state(torch.from_numpy(b.w.view(np.int32)).cuda())
            for s in range(4):
                a = orc.random_actions(1, s)
                ol, oi, oa, orew, odone, oinfo = (x.copy() for x in orc.step(a))
                gl, gi, ga, grew, gdone, ginfo = (x.cpu().numpy() for x in gpu.step(torch.from_numpy(a).cuda()))
                np.testing.assert_array_equal(gdone, odone); np.testing.assert_array_equal(ginfo, oinfo)
                np.testing.assert_allclose(grew, orew, rtol=1e-5, atol=1e-3)
                np.testing.assert_allclose(gl, ol, atol=1e-5); np.testing.assert_allclose(gi, oi, atol=1e-5)
                diff, imis = _compare_states(orc.get_state(), gpu.get_state().cpu().numpy().view(np.uint32), 1, cfg.n_drones)
                assert not imis.any() and diff.max() < 1e-4, (task, noise, s, diff.max())
            gpu.close(); orc.close()


def test_ground_contact_parity():
    """cfg.ground_contact (opt-in, parity with PyBullet unpinned: tests/test_oracle_ground.py): the kernel against the
    oracle on drones that are thrown at the plane.  2 048 evaluation envs, the wingman's z-velocity set to -3 m/s at random
    heights up to 0.6 m above the plane, then 6 free-running env-steps compared state by state (STATE_TOL 1e-4 after the first
    step from the identical state; 1e-3 along the rollout, as tests/test_gpu_parity.py::test_rollout_parity_from_reset)."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    from dronechase_amd import config as K, default_config
    from dronechase_amd.batched_env import BatchedEnv
    from oracle import te_oracle as O
    from tests._blob import Blob
    from tests.test_gpu_parity import _compare_states

    N = 2048
    cfg = default_config("evaluation", n_envs=N, motor_noise=1, seed=3, ground_contact=1, auto_reset=0)
    D = cfg.n_drones
    rest = cfg.ground_z + cfg.hull_half_height
    orc, gpu = O.OracleEnv(cfg, "f32", threads=8), BatchedEnv(cfg, "cuda:0")
    orc.reset()
    b = Blob(orc.get_state(), N, D)
    rng = np.random.default_rng(0)
    for e in range(N):
        z = rest + float(rng.uniform(0.01, 0.6))
        b.place(e, 0, (float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2)), z)); b.hover_ready(e, 0, cfg)
        b.set_f(e, 0, "VEL", [float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)), -3.0])
    orc.set_state(b.w); gpu.set_state(torch.from_numpy(b.w.view(np.int32)).cuda())
    a = np.zeros((N, 4), np.float32)
    touched = np.zeros(N, bool)
    for t in range(6):
        orc.step(a); gpu.step(torch.from_numpy(a).cuda())
        so, sg = orc.get_state(), gpu.get_state().cpu().numpy().view(np.uint32)
        diff, imis = _compare_states(so, sg, N, D)
        ok = orc.state_margins() > 1e-3
        assert not (imis & ok).any()
        assert diff[ok & ~imis].max() < (1e-4 if t == 0 else 1e-3), (t, diff[ok & ~imis].max())
        z = Blob(sg, N, D).dr[:, 0, K.D["POS"] + 2].view(np.float32)
        assert (z >= rest - 1e-5).all()
        touched |= z < rest + 1e-3
    assert touched.mean() > 0.1        # a share of the hulls sat on the plane at a step boundary (most touch it between two)
    from dronechase_amd import _lib
    with pytest.raises(_lib.TEError, match="level4 task family"):
        BatchedEnv(default_config("stage02", n_envs=64, ground_contact=1), "cuda:0")
    gpu.close(); orc.close()


@pytest.mark.parametrize("task", ["exp03", "level5"])
def test_drone_contact_parity(task):
    """cfg.drone_contact (opt-in, parity with PyBullet unpinned: tests/test_oracle_contact.py): engage_kernel against the oracle on
    pursuers that fly into each other: 1 024 envs, pursuers 0 and 1 a few centimetres apart with closing velocities, 5 free-running
    env-steps compared state by state."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    from dronechase_amd import config as K, default_config
    from dronechase_amd.batched_env import BatchedEnv
    from oracle import te_oracle as O
    from tests.test_gpu_parity import _compare_states

    N = 1024
    cfg = default_config(task, n_envs=N, motor_noise=1, seed=11, drone_contact=1, auto_reset=0)
    D = cfg.n_drones
    orc, gpu = O.OracleEnv(cfg, "f32", threads=8), BatchedEnv(cfg, "cuda:0")
    orc.reset()
    b = Blob(orc.get_state(), N, D)
    rng = np.random.default_rng(1)
    for e in range(N):
        c = rng.uniform(-1, 1, 3) + [0, 0, 3]
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        gap = rng.uniform(0.03, 0.25)
        for s, sign in ((0, -1.0), (1, 1.0)):
            b.place(e, s, c + sign * 0.5 * gap * d); b.hover_ready(e, s, cfg)
            b.set_f(e, s, "VEL", -sign * rng.uniform(0.0, 0.8) * d)
    orc.set_state(b.w); gpu.set_state(torch.from_numpy(b.w.view(np.int32)).cuda())
    touched = 0
    step_fn_o = orc.step_stacked if cfg.stacked_obs else orc.step
    step_fn_g = gpu.step_stacked if cfg.stacked_obs else gpu.step
    for t in range(5):
        a = orc.random_actions(3, t)
        step_fn_o(a); step_fn_g(torch.from_numpy(a).cuda())
        so, sg = orc.get_state(), gpu.get_state().cpu().numpy().view(np.uint32)
        n_words = N * (D * K.DRONE_WORDS + K.ENV_WORDS)
        diff, imis = _compare_states(so[:n_words], sg[:n_words], N, D)
        ok = orc.state_margins() > 1e-3
        assert not (imis & ok).any()
        assert diff[ok & ~imis].max() < (1e-4 if t == 0 else 2e-3), (t, diff[ok & ~imis].max())
        dr = Blob(sg[:n_words], N, D).dr
        gapn = np.linalg.norm(dr[:, 0, 0:3].view(np.float32) - dr[:, 1, 0:3].view(np.float32), axis=1)
        touched += int((np.abs(gapn - 2 * cfg.contact_radius) < 1e-4).sum())
    assert touched > N // 4       # many pairs sat exactly at the contact distance after a resolution
    from dronechase_amd import _lib
    with pytest.raises(_lib.TEError, match="engage_kernel"):
        BatchedEnv(default_config("stage02", n_envs=64, drone_contact=1), "cuda:0")
    gpu.close(); orc.close()
