"""Analytic known-answer tests of the oracle's L0 restatement (PyFlyt QuadX + Bullet free body).  These
numbers are NOT pinned against PyBullet (absent here; DESIGN.md "parity unpinned"): they pin the
restated algorithm against closed forms, and the Bullet math helpers against analytic rotations."""
import numpy as np
import pytest

from oracle import te_oracle as O

PREC = ["f64", "f32"]


def test_philox_known_answers():
    # Random123 kat_vectors: philox4x32-10
    assert list(O.philox([0, 0, 0, 0], [0, 0])) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert list(O.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2)) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert list(O.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0])) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
    # ... and philox4x32-7 (same file), the generator of the motor noise since round 3
    assert list(O.philox([0, 0, 0, 0], [0, 0], rounds=7)) == [0x5F6FB709, 0x0D893F64, 0x4F121F81, 0x4F730A48]
    assert list(O.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, rounds=7)) == [0x5207DDC2, 0x45165E59, 0x4D8EE751, 0x8C52F662]
    assert list(O.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0], rounds=7)) == \
        [0x4DFCCABA, 0x190A87F0, 0xC47362BA, 0xB6B5242A]


@pytest.mark.parametrize("prec", PREC)
def test_bullet_math_helpers(prec):
    tol = 1e-12 if prec == "f64" else 1e-6
    ident = [0, 0, 0, 1]
    np.testing.assert_allclose(O.vec_fn("ote_quat_to_mat", ident, 9, prec), np.eye(3).ravel(), atol=tol)
    # 90 deg about z (x,y,z,w order): x -> y
    qz = [0, 0, np.sin(np.pi / 4), np.cos(np.pi / 4)]
    np.testing.assert_allclose(O.rotate_vector(qz, [1, 0, 0], prec), [0, 1, 0], atol=tol)
    qx = [np.sin(np.pi / 4), 0, 0, np.cos(np.pi / 4)]
    np.testing.assert_allclose(O.rotate_vector(qx, [0, 1, 0], prec), [0, 0, 1], atol=tol)
    qy = [0, np.sin(np.pi / 4), 0, np.cos(np.pi / 4)]
    np.testing.assert_allclose(O.rotate_vector(qy, [0, 0, 1], prec), [1, 0, 0], atol=tol)
    # euler <-> quaternion round trip, ZYX composition R = Rz(yaw) Ry(pitch) Rx(roll)
    rng = np.random.RandomState(0)
    for _ in range(200):
        rpy = rng.uniform([-np.pi, -1.5, -np.pi], [np.pi, 1.5, np.pi])
        q = O.vec_fn("ote_quat_from_euler", rpy, 4, prec)
        assert abs(np.linalg.norm(q) - 1) < 10 * tol
        np.testing.assert_allclose(O.vec_fn("ote_euler_from_quat", q, 3, prec), rpy, atol=30 * tol)
        cr, sr, cp, sp, cy, sy = np.cos(rpy[0]), np.sin(rpy[0]), np.cos(rpy[1]), np.sin(rpy[1]), np.cos(rpy[2]), np.sin(rpy[2])
        Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]]); Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
        Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
        np.testing.assert_allclose(O.vec_fn("ote_quat_to_mat", q, 9, prec).reshape(3, 3), Rz @ Ry @ Rx, atol=30 * tol)
    # gimbal guard (pybullet getEulerFromQuaternion): pitch = +-pi/2 exactly
    q = O.vec_fn("ote_quat_from_euler", [0.0, np.pi / 2, 0.3], 4)
    e = O.vec_fn("ote_euler_from_quat", q, 3)
    assert e[0] == 0 and abs(e[1] - np.pi / 2) < 1e-12


def test_command_to_setpoint():
    # quadcopter.py:379-396: [vx, vy, 0, vz] = magnitude * unit(direction); zero direction stays zero
    np.testing.assert_allclose(O.vec_fn("ote_command_to_setpoint", [3, 0, 4, 0.5], 4), [0.3, 0, 0, 0.4], atol=1e-15)
    np.testing.assert_allclose(O.vec_fn("ote_command_to_setpoint", [0, 0, 0, 0.4], 4), [0, 0, 0, 0], atol=0)
    np.testing.assert_allclose(O.vec_fn("ote_command_to_setpoint", [0, -2, 0, 1], 4), [0, -1, 0, 0], atol=1e-15)


def test_spawn_sampler_ranges():
    # exp03_vFinal_task.py:584-608: r = 6 cap, theta in [0, pi) => y >= 0 ; phi in [acos(4/6), pi/2] => 0 <= z <= 4
    rng = np.random.RandomState(1)
    for _ in range(500):
        u, v = rng.rand(2)
        p = O.level4_position(6.0, 4.0, u, v)
        assert abs(np.linalg.norm(p) - 6) < 1e-12 and p[1] >= -1e-12 and -1e-12 <= p[2] <= 4 + 1e-12
        p = O.level4_position(2.0, 4.0, u, v)  # r < min_z: phi in [0, pi/2]
        assert abs(np.linalg.norm(p) - 2) < 1e-12 and p[2] >= -1e-12
    np.testing.assert_allclose(O.level4_position(6.0, 4.0, 0.0, 0.0), [6 * np.sin(np.arccos(4 / 6)), 0, 4.0], atol=1e-12)
    np.testing.assert_allclose(O.level4_position(6.0, 4.0, 0.5, 1.0 - 1e-15), [0, 6, 0], atol=1e-9)


@pytest.mark.parametrize("prec", PREC)
def test_free_fall_closed_form(prec):
    """Motors off (total_thrust = 0, no drag): semi-implicit Euler => v_n = -g n dt, z_n = z0 - g dt^2 n(n+1)/2."""
    cfg = O.default_config("exp03", quad__total_thrust=0.0, quad__drag_coef_xyz=0.0, quad__thrust_coef=1.0)
    n = 120
    pos, vel, eul, thr = O.fly(cfg, 6, [0, 0, 0, 0], n, [1, 2, 50], prec)
    dt, g = 1 / 240, 9.81
    k = np.arange(1, n + 1)
    tol = 1e-9 if prec == "f64" else 2e-4
    np.testing.assert_allclose(vel[:, 2], -g * dt * k, rtol=1e-6, atol=tol)
    np.testing.assert_allclose(pos[:, 2], 50 - g * dt * dt * k * (k + 1) / 2, atol=tol * 5)
    np.testing.assert_allclose(pos[:, :2], np.tile([1, 2], (n, 1)), atol=tol)
    np.testing.assert_allclose(eul, 0, atol=tol)


@pytest.mark.parametrize("prec", PREC)
def test_hover_equilibrium(prec):
    """Velocity set-point 0: the z-velocity integrator settles at throttle sqrt(m g / total_thrust)."""
    cfg = O.default_config("exp03")
    pos, vel, eul, thr = O.fly(cfg, 6, [0, 0, 0, 0], 240 * 8, [0, 0, 5], prec)
    hover = np.sqrt(cfg.quad.mass * cfg.quad.gravity / cfg.quad.total_thrust)
    np.testing.assert_allclose(thr[-1], hover, atol=2e-3)
    assert abs(vel[-1, 2]) < 5e-3 and np.abs(vel[-1, :2]).max() < 1e-6 and np.abs(eul[-1]).max() < 1e-6
    assert (thr >= 0).all() and (thr <= 1.0 + 1e-6).all()


def test_velocity_tracking_and_axis_conventions():
    cfg = O.default_config("exp03")
    # +x velocity => positive pitch (nose down, z-up frame); +y => negative roll
    pos, vel, eul, thr = O.fly(cfg, 6, [0.6, 0, 0, 0], 240 * 6, [0, 0, 5])
    assert abs(vel[-1, 0] - 0.6) < 0.02 and abs(vel[-1, 1]) < 1e-6
    assert eul[200:800, 1].max() > 0.02 and np.abs(eul[:, 0]).max() < 1e-9
    pos, vel, eul, thr = O.fly(cfg, 6, [0, 0.6, 0, 0], 240 * 6, [0, 0, 5])
    assert abs(vel[-1, 1] - 0.6) < 0.02 and eul[200:800, 0].min() < -0.02
    pos, vel, eul, thr = O.fly(cfg, 6, [0, 0, 0, 0.5], 240 * 6, [0, 0, 5])
    assert abs(vel[-1, 2] - 0.5) < 0.02
    # positive yaw-rate command => positive yaw rate (mix sign and reaction-torque sign consistent)
    pos, vel, eul, thr = O.fly(cfg, 6, [0, 0, 1.0, 0], 240 * 3, [0, 0, 5])
    yaw = np.unwrap(eul[:, 2])
    assert yaw[-1] > 1.0 and (np.diff(yaw[240:]) > 0).all()


def test_mode7_position_hold():
    """stage01 invader: mode 7 [x, y, r, z] holds its set-point (level2/components/quadcopter_manager.py:68)."""
    cfg = O.default_config("stage01")
    pos, vel, eul, thr = O.fly(cfg, 7, [0.5, -0.3, 0, 0.8], 240 * 12, [0.5, -0.3, 0.8])
    assert np.abs(pos[-1] - [0.5, -0.3, 0.8]).max() < 0.05
    pos, vel, eul, thr = O.fly(cfg, 7, [1.0, 0.0, 0, 1.0], 240 * 15, [0, 0, 1.0])
    assert np.abs(pos[-1] - [1.0, 0.0, 1.0]).max() < 0.05


def test_f32_build_tracks_f64():
    cfg = O.default_config("exp03")
    p64, v64, e64, t64 = O.fly(cfg, 6, [0.4, -0.2, 0, 0.1], 16 * 30, [0, 0, 3], "f64")
    p32, v32, e32, t32 = O.fly(cfg, 6, [0.4, -0.2, 0, 0.1], 16 * 30, [0, 0, 3], "f32")
    assert np.abs(p64 - p32).max() < 2e-4 and np.abs(v64 - v32).max() < 2e-4


def _fit_module():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("physics_fit", os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools", "physics_fit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_recorded_first_steps_against_the_two_quadrotor_presets():
    """tests/golden/ref_level5_obs.npz holds the IMU reads of seven wingmen at two consecutive env-steps right after a
    (re)spawn, each flying a constant BT command of 0.6 m/s (io_data0.h5, the only PyBullet + PyFlyt output in the
    reference tree).  tools/physics_fit.py flies the oracle through the same two steps with the hidden controller state of
    every wingman (z-velocity and linear-velocity integrators, which survive a respawn: quadcopter.py:433-478) fitted, and
    scores the 147 recorded numbers in units of their motor-noise scatter.  NOT a parity test — neither package
    can be run here — but the strongest constraint on the restated L0 there is:

      * the DEFAULT table of every task (round 3: te_config_default = TE_QUAD_CF2X_RECORDED_FIT, the recalled cf2x table with
        ang_vel_kp (roll, pitch) x 6 and motor_tau x 0.4 — the fewest changed entries that pass under the reference's loop as
        it reads) brings tilt, speed and rates to 1 within the scatter: chi^2 / dof < 1 and the best-fit bands are asserted ON
        THE DEFAULT;
      * the RECALLED table (PyFlyt's cf2x as recalled; te_quad_preset 0) reproduces every sign,
        near-hover thrust and zero yaw, but its attitude loop is 2-3x too slow (recorded / simulated tilt 1.8-3.2 after one
        step, chi^2 / dof 5.3): only upper bounds are asserted on it."""
    F = _fit_module()
    recalled = F.make_cfg(preset=0)
    default = O.default_config("level5", n_envs=1)
    assert default.quad_preset == 1 and abs(default.quad.motor_tau - 0.004) < 1e-9 and abs(default.quad.ang_vel_kp[0] - 0.048) < 1e-9
    assert recalled.quad_preset == 0 and abs(recalled.quad.motor_tau - 0.01) < 1e-9 and abs(recalled.quad.ang_vel_kp[0] - 0.008) < 1e-9
    sigma = F.noise_sigma(recalled, 150)       # one yardstick for every candidate: the recalled table's motor-noise scatter
    # signs and orders of magnitude with the recalled table (hidden state fitted)
    chi2, hid, sims = F.fit(recalled, sigma)
    for w in range(F.W):
        t, s = F.recorded(w), sims[w]
        for k in (0, 1, 3, 4):   # vx, vy, roll, pitch after one step: the command's horizontal direction shows up with the right sign
            if abs(t[k]) > 3 * sigma[k]:
                assert np.sign(s[k]) == np.sign(t[k]), (w, k)
        assert abs(t[5]) < 0.01 and abs(s[5]) < 0.01            # no yaw command, no yaw
        assert abs(s[2]) < 0.3 and abs(t[2]) < 0.3              # near-hover thrust from the first sub-steps
        assert 0.4 < hid[w, 0] < 0.75                           # the fitted z integrator sits near (below) the hover throttle 0.67
    r0 = F.ratios(sims)
    for name, hi in {"tilt1": 4.0, "tilt2": 2.5, "rate1": 3.0, "speed1": 5.0}.items():
        assert 0.7 < r0[name][0] and r0[name][1] < hi, (name, r0[name])
    assert chi2 / F.DOF > 3.0                                   # ... and it fails the recording: why it is no longer the default
    # the default of every task
    default.control_every_substep = 1
    fit = F.row("te_config_default", default, sigma)
    assert fit["chi2_dof"] < 1.0 and fit["chi2_dof"] < 0.2 * chi2 / F.DOF, (fit["chi2_dof"], chi2 / F.DOF)
    for name, (lo, hi) in {"tilt1": (0.9, 1.3), "tilt2": (0.9, 1.25), "rate1": (0.7, 1.5), "speed1": (0.7, 1.2)}.items():
        assert lo < fit["ratios"][name][0] and fit["ratios"][name][1] < hi, (name, fit["ratios"][name])
    for w in range(F.W):
        assert 0.4 < fit["zv_i"][w] < 0.75
    # The second yardstick (round-3 review): each table scored in units of ITS OWN motor-noise scatter (the default flies calmer, its scatter
    # is 4-5 x smaller).  There the default does NOT pass either (chi^2 / dof 2.8 at the end of round 3; the recalled table 5.3), and other
    # candidates of tools/physics_fit.py fit as well or better (120 Hz control + ang_pos_kp x 2 + total_thrust x 2: 0.20 / 0.29): the recording
    # does not identify the entries, the default is ONE of several tables it supports, and PyBullet parity stays unpinned.
    own = F.row("te_config_default, own noise", default, F.noise_sigma(default, 150))
    assert 1.0 < own["chi2_dof"] < 5.0, own["chi2_dof"]
    assert own["chi2_dof"] < chi2 / F.DOF               # still the better of the two presets under either yardstick


@pytest.mark.parametrize("preset", [1, 0])
def test_both_presets_fly(preset):
    """te_quad_preset(TE_QUAD_CF2X_RECORDED_FIT) — the default — and the recalled table: hover equilibrium, velocity tracking and the
    mode-7 hold; with the default's stiffer rate loop and near-instant motors (dt / tau = 1.04) the discrete lag must not ring."""
    cfg = O.default_config("exp03", quad_preset=preset)
    if preset == 0:
        assert cfg.quad_preset == 0 and abs(cfg.quad.motor_tau - 0.01) < 1e-9 and abs(cfg.quad.ang_vel_kp[0] - 0.008) < 1e-9
    else:
        assert O.default_config("exp03").quad.motor_tau == cfg.quad.motor_tau and O.default_config("exp03").quad_preset == 1
        assert cfg.quad_preset == 1 and abs(cfg.quad.motor_tau - 0.004) < 1e-9 and abs(cfg.quad.ang_vel_kp[0] - 0.048) < 1e-9
    assert abs(cfg.quad.ang_vel_kp[2] - 0.01) < 1e-9            # yaw untouched
    pos, vel, eul, thr = O.fly(cfg, 6, [0, 0, 0, 0], 240 * 8, [0, 0, 5])
    hover = np.sqrt(cfg.quad.mass * cfg.quad.gravity / cfg.quad.total_thrust)
    np.testing.assert_allclose(thr[-1], hover, atol=2e-3)
    assert abs(vel[-1, 2]) < 5e-3 and np.abs(eul).max() < 1e-6
    pos, vel, eul, thr = O.fly(cfg, 6, [0.6, 0, 0, 0.2], 240 * 6, [0, 0, 5])
    assert abs(vel[-1, 0] - 0.6) < 0.02 and abs(vel[-1, 2] - 0.2) < 0.02 and np.abs(eul).max() < 0.5
    cfg7 = O.default_config("stage01", quad_preset=preset)
    pos, vel, eul, thr = O.fly(cfg7, 7, [1.0, 0.0, 0, 1.0], 240 * 15, [0, 0, 1.0])
    assert np.abs(pos[-1] - [1.0, 0.0, 1.0]).max() < 0.05


def test_control_every_substep_switch():
    """cfg.control_every_substep = 0: QuadX.update_control on every second physics sub-step (PyFlyt's own 120 Hz), the motors
    keep the last pwm in between.  Same equilibrium, different transient."""
    a = O.default_config("exp03")
    b = O.default_config("exp03", control_every_substep=0)
    pa, va, ea, ta = O.fly(a, 6, [0.5, 0.2, 0, 0.1], 240 * 6, [0, 0, 5])
    pb, vb, eb, tb = O.fly(b, 6, [0.5, 0.2, 0, 0.1], 240 * 6, [0, 0, 5])
    assert np.abs(va[-1] - [0.5, 0.2, 0.1]).max() < 0.03 and np.abs(vb[-1] - [0.5, 0.2, 0.1]).max() < 0.05
    assert np.abs(va[:240] - vb[:240]).max() > 1e-3     # the transient differs: the integrators run at half the rate
    # at 120 Hz the pwm of an odd sub-step is the previous one's: throttle changes twice as rarely
    assert np.abs(tb[:40] - ta[:40]).max() > 1e-4
