"""Parity tests proper: the HIP path (through the C ABI, dronechase_amd/libthreatengage.so) against the CPU
oracle on identical seeded inputs.  Run on an MI355X: `pytest -m gpu`.

Tolerances (written here, justified in DESIGN.md "parity"):
  STATE_TOL  1e-4  max |delta| of any float word of the drone/env state after ONE env.step (16 physics
                   sub-steps) started from an identical state — BASELINE.json's per-step tolerance
  OBS_TOL    1e-5  normalised observations; REWARD_TOL 1e-3 relative-ish (rewards reach +-1000)
  integers (armed flags, munition, counters, done, info) must match exactly,
except in environments the oracle flags as AMBIGUOUS: some discrete decision of that step had
|value - threshold| < MARGIN (a distance within 1e-4 m of a range, etc.), where float32 rounding may
legitimately flip the branch.  Ambiguous envs must stay a tiny fraction and every mismatch must be one.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STATE_TOL = 1e-4
OBS_TOL = 1e-5
MARGIN = 1e-4
TASKS = [("exp03", {}), ("exp02", {}), ("exp04", {}), ("exp05", {}), ("stage02", {}), ("stage02", {"n_invaders": 8}), ("stage01", {}),
         # SURVEY.md A.7 switches: PyFlyt-native 120 Hz controller (control_every_substep = 0), the recalled quadrotor table (the default of
         # every task is the recorded-fit table since round 3)
         ("exp03", {"control_every_substep": 0}), ("stage01", {"control_every_substep": 0}), ("exp03", {"quad_preset": 0}),
         ("exp03", {"lidar_channels": 2}), ("stage02", {"lidar_channels": 2})]


def _gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box (no CPU fallback exists)")
    return torch


def _split(w, N, D):
    from dronechase_amd import config as K
    dr = w[: N * D * K.DRONE_WORDS].reshape(N, D, K.DRONE_WORDS)
    er = w[N * D * K.DRONE_WORDS:].reshape(N, K.ENV_WORDS)
    return dr, er


def _float_words():
    from dronechase_amd import config as K
    fd = [i for i in range(K.DRONE_WORDS) if i not in K.D_INT_WORDS]
    fe = [K.E["LAST_DIST"], *range(K.E["LAST_ACTION"], K.E["LAST_ACTION"] + 4), K.E["PREV_SNAP_MIN"]]
    return fd, list(K.D_INT_WORDS), fe, list(K.E_INT_WORDS)


def _compare_states(so, sg, N, D, rate_error_tol=STATE_TOL):
    """-> (per-env max float diff in units of STATE_TOL-equivalents, per-env bool any-int-mismatch).  `rate_error_tol`: the bound of the three
    PID_AV_E words (the rate loop's previous error), scaled so that every word compares against STATE_TOL."""
    fd, idw, fe, iew = _float_words()
    do, eo = _split(so, N, D)
    dg, eg = _split(sg, N, D)
    from dronechase_amd import config as K
    # EVERY float word is held to STATE_TOL, the controller's memories included.  Measured per word group on the MI355X (tools/parity_margins.py,
    # profiles/r04_e_parity_margins.txt; 5 task variants x motor noise on / off x 8 checkpoints x 2 048 envs): pose / velocities <= 5.2e-5,
    # throttles <= 9.3e-6, the previous rate error PID_AV_E <= 8.7e-5 (control_every_substep = 0; 4.3e-5 under either quadrotor table with the
    # reference's loop), every other PID word <= 1.1e-6, IMU reads <= 3.6e-5.  (Round 3 had relaxed the PID words to 2 x STATE_TOL when the
    # recorded-fit table became the default; the measurement shows that no word needs it under the reference's loop.)  The one exception is
    # cfg.control_every_substep = 0 (PyFlyt's own 120 Hz controller, not a reference mode): there the previous rate error reaches 8.7e-5 ... 1.0e-4
    # depending on the build's rounding (it is the difference of two rad/s quantities held for two sub-steps), so that variant passes rate_error_tol = 2e-4.
    scale = np.array([STATE_TOL / rate_error_tol if K.D["PID_AV_E"] <= w < K.D["PID_AV_E"] + 3 else 1.0 for w in fd])
    fdiff = (np.abs(do[..., fd].view(np.float32).astype(np.float64) - dg[..., fd].view(np.float32)) * scale).reshape(N, -1).max(1)
    ediff = np.abs(eo[:, fe].view(np.float32).astype(np.float64) - eg[:, fe].view(np.float32)).max(1)
    imis = (do[..., idw] != dg[..., idw]).any(axis=(1, 2)) | (eo[:, iew] != eg[:, iew]).any(axis=1)
    return np.maximum(fdiff, ediff), imis


@pytest.mark.parametrize("task,over", TASKS)
@pytest.mark.parametrize("noise", [0, 1])
def test_single_step_parity(task, over, noise):
    """Oracle rolls an episode forward; at several checkpoints its state blob is loaded into the GPU env
    (te_set_state) and both take ONE step with the same actions."""
    torch = _gpu()
    from dronechase_amd import default_config
    from dronechase_amd.batched_env import BatchedEnv
    from oracle import te_oracle as O

    N = 2048
    cfg = default_config(task, n_envs=N, motor_noise=noise, seed=17, **over)
    D = cfg.n_drones
    orc = O.OracleEnv(cfg, "f32", threads=8)
    gpu = BatchedEnv(cfg, "cuda:0")
    orc.reset(); gpu.reset()
    # identical reset (same Philox stream; trig differs by ulps)
    diff, imis = _compare_states(orc.get_state(), gpu.get_state().cpu().numpy().view(np.uint32), N, D)
    assert not imis.any() and diff.max() < 1e-5
    gl, gi, ga = (x.cpu().numpy() for x in gpu.observe())
    ol, oi, oa = orc.observe()
    assert (gl == 1.0).all() and (ol == 1.0).all()  # empty sphere right after reset
    np.testing.assert_allclose(gi, oi, atol=OBS_TOL)
    step = 0
    n_ambiguous = 0
    for chk in range(8):
        for _ in range(37):
            orc.step(orc.random_actions(23, step)); step += 1
        gpu.set_state(torch.from_numpy(orc.get_state().view(np.int32)).cuda())
        a = orc.random_actions(23, step); step += 1
        ol, oi, oa, orew, odone, oinfo = (x.copy() for x in orc.step(a))
        otl, oti = orc.t_lidar.copy(), orc.t_inertial.copy()
        ok = orc.margins() > MARGIN
        gl, gi, ga, grew, gdone, ginfo = (x.cpu().numpy() for x in gpu.step(torch.from_numpy(a).cuda()))
        gtl, gti = gpu.t_lidar.cpu().numpy(), gpu.t_inertial.cpu().numpy()
        diff, imis = _compare_states(orc.get_state(), gpu.get_state().cpu().numpy().view(np.uint32), N, D,
                                     rate_error_tol=2e-4 if over.get("control_every_substep") == 0 else STATE_TOL)
        n_ambiguous += int((~ok).sum())
        # every discrete mismatch must be an ambiguous env
        assert not (imis & ok).any(), f"{task}: integer state mismatch outside ambiguous envs at step {step}"
        assert not ((odone != gdone) & ok).any()
        assert not ((oinfo != ginfo).any(1) & ok).any()
        good = ok & ~imis
        assert diff[good].max() < STATE_TOL, f"{task}: state diff {diff[good].max():.3e}"
        np.testing.assert_allclose(grew[good], orew[good], rtol=1e-5, atol=1e-3)
        np.testing.assert_allclose(gi[good], oi[good], atol=OBS_TOL)
        np.testing.assert_allclose(ga[good], oa[good], atol=0)
        lid_bad = np.abs(gl - ol).reshape(N, -1).max(1) > OBS_TOL
        # a LIDAR cell index may flip when an angle sits within float rounding of a cell edge
        assert (lid_bad & good).sum() <= max(2, N // 500), f"{task}: {int((lid_bad & good).sum())} LIDAR tiles differ"
        d = odone.astype(bool) & gdone.astype(bool) & good
        if d.any():  # terminal observations of auto-reset envs
            np.testing.assert_allclose(gti[d], oti[d], atol=OBS_TOL)
            assert (np.abs(gtl[d] - otl[d]).reshape(int(d.sum()), -1).max(1) > OBS_TOL).sum() <= 1
            assert (gl[d] == 1.0).all()  # reset observation: empty sphere
    assert n_ambiguous <= 8 * N // 50  # ambiguous decisions stay rare (< 2 % of env-steps)
    gpu.close(); orc.close()


@pytest.mark.parametrize("task", ["exp03", "stage01", "stage02"])
def test_rollout_parity_from_reset(task):
    """Free-running rollout from the same seeds (no re-synchronisation): per-step rewards/dones agree for
    the envs that have not yet hit an ambiguous decision, and the state error stays bounded there."""
    torch = _gpu()
    from dronechase_amd import default_config
    from dronechase_amd.batched_env import BatchedEnv
    from oracle import te_oracle as O

    N, T = 512, 120
    cfg = default_config(task, n_envs=N, motor_noise=1, seed=5)
    D = cfg.n_drones
    orc = O.OracleEnv(cfg, "f32", threads=8)
    gpu = BatchedEnv(cfg, "cuda:0")
    orc.reset(); gpu.reset()
    clean = np.ones(N, bool)  # envs whose state-changing decisions were never near a threshold so far
    for s in range(T):
        a = orc.random_actions(7, s)
        ol, oi, oa, orew, odone, oinfo = orc.step(a)
        gl, gi, ga, grew, gdone, ginfo = (x.cpu().numpy() for x in gpu.step(torch.from_numpy(a).cuda()))
        clean &= orc.state_margins() > 1e-3  # generous: errors accumulate along the rollout
        assert not ((odone != gdone) & clean).any(), f"step {s}"
        rew_ok = clean & (orc.margins() > 1e-3)  # reward-only thresholds (e.g. the 0.01 m approach bonus)
        np.testing.assert_allclose(grew[rew_ok], orew[rew_ok], rtol=1e-4, atol=2e-2)
        np.testing.assert_allclose(gi[clean], oi[clean], atol=2e-4)
    diff, imis = _compare_states(orc.get_state(), gpu.get_state().cpu().numpy().view(np.uint32), N, D)
    # measured on the MI355X (tools/parity_margins.py): 0.963 (exp03), 0.992 (stage01), 0.996 (stage02) of the envs never come within 1e-3 of a
    # state-changing threshold in 120 steps; the bound leaves 3 points of margin for another seed, not half of the envs
    print(f"{task}: clean fraction {clean.mean():.3f}")
    assert clean.mean() > {"exp03": 0.93, "stage01": 0.96, "stage02": 0.96}[task]
    assert not (imis & clean).any()
    assert diff[clean].max() < 5e-3  # 120 env-steps = 1920 sub-steps of float32 drift, closed-loop
    gpu.close(); orc.close()


def test_against_float64_oracle():
    """GPU float32 vs the float64 build of the oracle (the reference computes in double): one step from a
    common state."""
    torch = _gpu()
    from dronechase_amd import default_config
    from dronechase_amd.batched_env import BatchedEnv
    from oracle import te_oracle as O

    N = 1024
    cfg = default_config("exp03", n_envs=N, motor_noise=1, seed=2)
    orc = O.OracleEnv(cfg, "f64", threads=8)
    gpu = BatchedEnv(cfg, "cuda:0")
    orc.reset()
    for s in range(60):
        orc.step(orc.random_actions(1, s))
    gpu.set_state(torch.from_numpy(orc.get_state().view(np.int32)).cuda())
    orc.set_state(orc.get_state())  # round the oracle's state to float32 too: common starting point
    a = orc.random_actions(1, 60)
    _, oi, _, orew, odone, _ = orc.step(a)
    ok = orc.margins() > MARGIN
    _, gi, _, grew, gdone, _ = (x.cpu().numpy() for x in gpu.step(torch.from_numpy(a).cuda()))
    diff, imis = _compare_states(orc.get_state(), gpu.get_state().cpu().numpy().view(np.uint32), N, cfg.n_drones)
    good = ok & ~imis
    assert not (imis & ok).any() and not ((odone != gdone) & ok).any()
    assert diff[good].max() < STATE_TOL
    np.testing.assert_allclose(gi[good], oi[good], atol=OBS_TOL)
    np.testing.assert_allclose(grew[good], orew[good], rtol=1e-5, atol=1e-3)
    gpu.close(); orc.close()


def test_ragged_env_counts():
    """n_envs that are not multiples of the 64-env tile / 256-thread block: 1, 63, 65, 257."""
    torch = _gpu()
    from dronechase_amd import default_config
    from dronechase_amd.batched_env import BatchedEnv
    from oracle import te_oracle as O

    for N in (1, 63, 65, 257):
        cfg = default_config("exp03", n_envs=N, motor_noise=0, seed=9)
        orc = O.OracleEnv(cfg, "f32")
        gpu = BatchedEnv(cfg, "cuda:0")
        orc.reset(); gpu.reset()
        for s in range(5):
            a = orc.random_actions(3, s)
            ol, oi, oa, orew, odone, oinfo = orc.step(a)
            gl, gi, ga, grew, gdone, ginfo = (x.cpu().numpy() for x in gpu.step(torch.from_numpy(a).cuda()))
            np.testing.assert_array_equal(gdone, odone)
            np.testing.assert_allclose(gl, ol, atol=OBS_TOL)
            np.testing.assert_allclose(gi, oi, atol=OBS_TOL)
            np.testing.assert_allclose(grew, orew, atol=1e-3)
        gpu.close(); orc.close()


def test_ragged_env_counts_of_the_stacked_tasks():
    """The same for te_step_stacked / te_observe_stacked: level5 (18 drones), level5_c1 (12) and level5_fusion (36: the 64-bit masks and
    agent_rows_kernel) with 1, 63 and 65 envs."""
    torch = _gpu()
    from dronechase_amd import default_config
    from dronechase_amd.batched_env import BatchedEnv
    from oracle import te_oracle as O

    for task in ("level5", "level5_c1", "level5_fusion"):
        for N in (1, 63, 65):
            cfg = default_config(task, n_envs=N, motor_noise=0, seed=9)
            orc = O.OracleEnv(cfg, "f32")
            gpu = BatchedEnv(cfg, "cuda:0")
            orc.reset(); gpu.reset()
            gs, gm, gi, ga = (x.cpu().numpy() for x in gpu.observe_stacked())
            os_, om, oi, oa = orc.observe_stacked()
            np.testing.assert_array_equal(gm, om); np.testing.assert_allclose(gs, os_, atol=OBS_TOL); np.testing.assert_allclose(gi, oi, atol=OBS_TOL)
            for s in range(4):
                a = orc.random_actions(3, s)
                os_, om, oi, oa, orew, odone, oinfo = orc.step_stacked(a)
                ok = orc.stack_margins() > 5e-5       # cell decisions inside the float tolerance are not comparable
                gs, gm, gi, ga, grew, gdone, ginfo = (x.cpu().numpy() for x in gpu.step_stacked(torch.from_numpy(a).cuda()))
                np.testing.assert_array_equal(gdone, odone); np.testing.assert_array_equal(ginfo, oinfo)
                np.testing.assert_array_equal(gm[ok], om[ok])
                np.testing.assert_allclose(gs[ok], os_[ok], atol=OBS_TOL)
                np.testing.assert_allclose(gi, oi, atol=OBS_TOL)
                np.testing.assert_allclose(grew, orew, atol=2e-3)
            gpu.close(); orc.close()


def test_random_actions_match_oracle():
    torch = _gpu()
    from dronechase_amd import default_config
    from dronechase_amd.batched_env import BatchedEnv
    from oracle import te_oracle as O

    cfg = default_config("exp03", n_envs=1000, env_index_base=123456789012, seed=1)
    orc = O.OracleEnv(cfg, "f32")
    gpu = BatchedEnv(cfg, "cuda:0")
    for step in (0, 1, 2 ** 33 + 5):
        np.testing.assert_array_equal(gpu.random_actions(1234, step).cpu().numpy(), orc.random_actions(1234, step))
    gpu.close(); orc.close()


def test_state_blob_roundtrip():
    torch = _gpu()
    from dronechase_amd import default_config
    from dronechase_amd.batched_env import BatchedEnv

    cfg = default_config("stage02", n_envs=300, seed=4)
    gpu = BatchedEnv(cfg, "cuda:0")
    gpu.reset()
    for s in range(3):
        gpu.step(gpu.random_actions(1, s))
    w = gpu.get_state()
    gpu2 = BatchedEnv(cfg, "cuda:0")
    gpu2.set_state(w)
    assert torch.equal(gpu2.get_state(), w)
    a = gpu.random_actions(1, 3)
    o1 = [x.clone() for x in gpu.step(a)]
    o2 = [x.clone() for x in gpu2.step(a)]
    for x, y in zip(o1, o2):
        assert torch.equal(x, y)  # checkpoint/resume is bit-exact
    gpu.close(); gpu2.close()


def test_c_abi_error_paths():
    torch = _gpu()
    import ctypes as C
    from dronechase_amd import _lib, default_config, TEError
    from dronechase_amd.batched_env import BatchedEnv

    L = _lib.load()
    bad = default_config("exp03", n_envs=8)
    bad.struct_size = 12
    h = C.c_void_p()
    assert L.te_create(C.byref(bad), 0, C.byref(h)) != 0 and b"struct_size" in L.te_last_error()
    bad = default_config("exp03", n_envs=0)
    assert L.te_create(C.byref(bad), 0, C.byref(h)) != 0
    bad = default_config("exp03", n_envs=8, n_invaders=40)
    assert L.te_create(C.byref(bad), 0, C.byref(h)) != 0
    assert L.te_create(C.byref(default_config("exp03", n_envs=8)), 99, C.byref(h)) != 0 and b"device_id" in L.te_last_error()
    env = BatchedEnv(default_config("exp03", n_envs=8), "cuda:0")
    with pytest.raises(ValueError):
        env.step(torch.zeros((7, 4), device="cuda:0"))
    with pytest.raises(ValueError):
        env.set_state(torch.zeros(5, dtype=torch.int32, device="cuda:0"))
    buf = torch.zeros(8 * 4 + 1, device="cuda:0")
    rc = L.te_random_actions(env._h, C.c_void_p(buf.data_ptr() + 4), 0, 0, None)
    assert rc != 0 and b"aligned" in L.te_last_error()
    w = torch.zeros(3, dtype=torch.int32, device="cuda:0")
    assert L.te_get_state(env._h, C.c_void_p(w.data_ptr()), 3, None) != 0
    env.close()
