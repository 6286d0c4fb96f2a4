"""Opt-in ground plane (cfg.ground_contact): the reference loads plane.urdf at z = -6
(entities_manager.py:120-124, immovable_structures.py:123-135) and lets Bullet resolve the contact.  PARITY UNPINNED: neither
pybullet nor the cf2x collision geometry is in the tree; the model (inelastic normal contact at hull_half_height above the
plane, Coulomb friction 0.5, no contact torque) is a stated approximation, off in every preset.  These tests hold the
oracle to that statement; tests/test_gpu_scenarios.py::test_ground_contact_parity holds the kernel to the oracle."""
import numpy as np

from oracle import te_oracle as O
from tests.test_oracle_tasks import arena, load, step


def _env(**over):
    cfg = O.default_config("evaluation", n_envs=1, motor_noise=0, auto_reset=0, **over)
    env = O.OracleEnv(cfg, "f64")
    env.reset()
    return cfg, env


def test_presets_leave_the_ground_off():
    for task in ("stage01", "stage02", "exp02", "exp03", "exp04", "exp05", "evaluation", "level5"):
        c = O.default_config(task)
        assert c.ground_contact == 0 and c.ground_z == -6.0 and abs(c.hull_half_height - 0.0125) < 1e-9


def _dive(ground):
    """A wingman 0.29 m above the plane falling at 4 m/s (more than its motors can arrest in that distance), its target
    level with it and far away: z of the hull centre over the next 12 env-steps."""
    cfg, env = _env(ground_contact=ground)
    rest = cfg.ground_z + cfg.hull_half_height
    arena(cfg, env, agent=(0, 0, rest + 0.29), invaders=((9.0, 0, rest + 0.29),))
    b = load(env, cfg)
    b.set_f(0, 0, "VEL", [0.0, 0.0, -4.0])
    env.set_state(b.w)
    zs = []
    for _ in range(12):
        step(env)
        zs.append(float(load(env, cfg).f(0, 0, "POS", 3)[2]))
    return cfg, env, rest, np.array(zs)


def test_a_drone_flown_into_the_ground_stops_on_it_and_can_take_off_again():
    cfg, env, rest, zs = _dive(1)
    assert rest - 1e-6 <= zs.min() < rest + 2e-2      # it reached the plane (sampled at env-step boundaries: it may have left it again by up to 16 sub-steps of climb) and never went below
    _, _, _, zs0 = _dive(0)
    assert zs0.min() < rest - 0.02                                       # without the plane the same dive goes through z = rest
    # the contact is one-sided: with its target 3 m up the wingman leaves the ground again
    b = load(env, cfg)
    b.place(0, cfg.n_pursuers, (3.0, 0, rest + 3.0)); b.hover_ready(0, cfg.n_pursuers, cfg)
    b.refresh_snapshot(0)
    env.set_state(b.w)
    for _ in range(60):
        step(env)
    assert load(env, cfg).f(0, 0, "POS", 3)[2] > rest + 0.2


def test_friction_takes_the_tangential_speed_of_an_impact():
    """One physics sub-step on a hull that hits the plane with v = (1, 0, -2): v_z -> 0, |v_t| reduced by 0.5 * 2 = 1."""
    cfg, env = _env(ground_contact=1, substeps=1)
    rest = cfg.ground_z + cfg.hull_half_height
    arena(cfg, env, agent=(0, 0, rest + 1e-4), invaders=((0, 0, -9.0),))
    b = load(env, cfg)
    b.set_f(0, 0, "VEL", [1.0, 0.0, -2.0])
    env.set_state(b.w)
    step(env)
    b = load(env, cfg)
    v = b.f(0, 0, "VEL", 3)
    assert abs(b.f(0, 0, "POS", 3)[2] - rest) < 1e-9 and v[2] == 0.0
    assert 0.0 <= v[0] < 0.05 and abs(v[1]) < 1e-6          # (1 - 0.5 * (2 + g dt + ...)) clipped at 0
