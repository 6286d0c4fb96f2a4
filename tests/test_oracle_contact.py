"""cfg.drone_contact (opt-in, parity with PyBullet unpinned: DESIGN.md): the stated sphere model in the oracle."""
import numpy as np

from tests._blob import Blob


def _env(**over):
    from oracle import te_oracle as O
    cfg = O.default_config("exp03", n_envs=1, motor_noise=0, auto_reset=0, substeps=0, observe_lag=0, shoot_range=0.0, explosion_range=0.0, **over)
    orc = O.OracleEnv(cfg, "f64")
    b = Blob(np.zeros(orc.state_words(), np.uint32), 1, cfg.n_drones)
    for s in range(cfg.n_drones):
        b.place(0, s, (30.0 + s, 0, 0), armed=0)
    b.set_ei(0, "ROUND", 9); b.set_ei(0, "MAX_STEP", 10 ** 6); b.set_ei(0, "EPISODE", 1)
    return cfg, orc, b


def test_head_on_pair_is_separated_and_shares_its_normal_velocity():
    cfg, orc, b = _env(drone_contact=1)
    b.place(0, 0, (0, 0, 3)); b.place(0, 1, (0.08, 0, 3)); b.place(0, 2, (5, 5, 3))
    b.set_f(0, 0, "VEL", [0.5, 0.1, 0]); b.set_f(0, 1, "VEL", [-0.3, 0, 0.2])
    b.refresh_snapshot(0)
    orc.set_state(b.w); orc.step(np.zeros((1, 4), np.float32))
    a = Blob(orc.get_state(), 1, cfg.n_drones)
    p0, p1 = a.f(0, 0, "POS", 3), a.f(0, 1, "POS", 3)
    np.testing.assert_allclose(np.linalg.norm(p1 - p0), 2 * cfg.contact_radius, atol=1e-6)      # 0.12 m apart now
    np.testing.assert_allclose(p0 + p1, [0.08, 0, 6], atol=1e-6)                                  # moved symmetrically
    v0, v1 = a.f(0, 0, "VEL", 3), a.f(0, 1, "VEL", 3)
    np.testing.assert_allclose([v0[0], v1[0]], [0.1, 0.1], atol=1e-6)                             # the mean of 0.5 and -0.3: inelastic, equal masses
    np.testing.assert_allclose([v0[1], v0[2], v1[1], v1[2]], [0.1, 0, 0, 0.2], atol=1e-6)         # tangential components untouched
    np.testing.assert_allclose(a.f(0, 0, "OBS_POS", 3), [0, 0, 3], atol=0)                        # this step's IMU read is not touched


def test_separating_pair_keeps_its_velocity_and_disarmed_drones_do_not_collide():
    cfg, orc, b = _env(drone_contact=1)
    b.place(0, 0, (0, 0, 3)); b.place(0, 1, (0.08, 0, 3)); b.place(0, 2, (0.04, 0.02, 3), armed=0); b.place(0, 3, (5, 5, 3))
    b.set_f(0, 0, "VEL", [-0.5, 0, 0]); b.set_f(0, 1, "VEL", [0.3, 0, 0])
    b.refresh_snapshot(0)
    orc.set_state(b.w); orc.step(np.zeros((1, 4), np.float32))
    a = Blob(orc.get_state(), 1, cfg.n_drones)
    np.testing.assert_allclose(a.f(0, 0, "VEL", 3), [-0.5, 0, 0], atol=0); np.testing.assert_allclose(a.f(0, 1, "VEL", 3), [0.3, 0, 0], atol=0)
    np.testing.assert_allclose(a.f(0, 2, "POS", 3), [0.04, 0.02, 3], atol=1e-7)
    # the switch is off in every preset
    cfg0, orc0, b0 = _env()
    assert cfg0.drone_contact == 0
    b0.place(0, 0, (0, 0, 3)); b0.place(0, 1, (0.08, 0, 3)); b0.place(0, 2, (5, 5, 3)); b0.refresh_snapshot(0)
    orc0.set_state(b0.w); orc0.step(np.zeros((1, 4), np.float32))
    np.testing.assert_allclose(Blob(orc0.get_state(), 1, cfg0.n_drones).f(0, 1, "POS", 3), [0.08, 0, 3], atol=1e-7)
