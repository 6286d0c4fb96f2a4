"""Level5DumbMultiObs (threatsense/level5/level5_dumb_multiobs.py + tasks/level5_dumb_multiobject_task.py): the imitation-data collector's
environment — 7 wingmen all flown by the behaviour tree, 30 invader slots (37 drones per env: 64-bit slot masks), the student observation
of every wingman.  The oracle's rules here; tests/test_gpu_level5_dumb.py holds the HIP path to the oracle."""
import numpy as np

from tests._blob import Blob


def _env(n=8, **over):
    from oracle import te_oracle as O
    cfg = O.default_config("level5_dumb", n_envs=n, **over)
    return cfg, O.OracleEnv(cfg, "f32")


def test_constants_are_the_references():
    cfg, o = _env(1)
    # level5_dumb_multiobject_task.py:82-100: 6 + 1 pursuers, 30 invader slots, ceil((30 - 5) / 1 + 1) = 26 rounds, (5 + 30) * 26 // 2 = 455 rounds of munition
    assert (cfg.n_pursuers, cfg.n_invaders, cfg.n_rounds, cfg.munition) == (7, 30, 26, 455)
    assert (cfg.initial_invaders, cfg.invaders_per_round, cfg.agent_scripted, cfg.reward_model, cfg.agent_death_terminates, cfg.stacked_obs) == (5, 1, 1, 1, 0, 1)
    assert cfg.n_drones == 37


def test_rounds_arm_four_plus_round_invaders_and_every_wingman_is_scripted():
    cfg, o = _env(4, seed=2, motor_noise=0)
    o.reset()
    P = cfg.n_pursuers
    b = Blob(o.get_state(), 4, 37)
    for e in range(4):
        armed = [b.i(e, s, "ARMED") for s in range(37)]
        assert armed[:P] == [1] * P and armed[P:P + 5] == [1] * 5 and sum(armed[P + 5:]) == 0      # round 1: 5 invaders (:173-184)
        assert b.ei(e, "SNAP_MASK") == (1 << 12) - 1 and b.ei(e, "SNAP_MASK_HI") == 0
    # a later round reaches slots beyond bit 31: load round 25 cleared -> round 26 arms min(25 + 5, 30) = 30 invaders
    for e in range(4):
        for j in range(30):
            b.set_i(e, P + j, "ARMED", 0)
        b.set_ei(e, "ROUND", 25); b.set_ei(e, "SNAP_MASK", (1 << P) - 1); b.set_ei(e, "SNAP_MASK_HI", 0)
    o.set_state(b.w)
    st, m, inert, la, act, r, d, info = o.step_students()
    a = Blob(o.get_state(), 4, 37)
    for e in range(4):
        assert a.ei(e, "ROUND") == 26 and sum(a.i(e, P + j, "ARMED") for j in range(30)) == 30
        assert a.ei(e, "SNAP_MASK") & 0xFFFFFFFF == 0xFFFFFFFF and a.ei(e, "SNAP_MASK_HI") == (1 << 5) - 1         # 37 bits
    # every pursuer, the agent included, obeys the behaviour tree: a 0.6 m/s set-point, and the teacher's action is (unit direction, 0.6)
    st, m, inert, la, act, r, d, info = o.step_students()
    assert act.all()
    np.testing.assert_allclose(np.linalg.norm(la[..., :3], axis=-1), 1.0, atol=1e-5)
    np.testing.assert_allclose(la[..., 3], 0.6, atol=1e-7)
    a = Blob(o.get_state(), 4, 37)
    for e in range(4):
        for p in range(P):
            sp = a.f(e, p, "SETPOINT", 4)
            np.testing.assert_allclose(np.linalg.norm(sp[[0, 1, 3]]), 0.6, atol=1e-5)
            np.testing.assert_allclose(sp[[0, 1, 3]] / 0.6, la[e, p, :3], atol=1e-5)


def test_the_episode_survives_the_agent_and_the_reward_is_the_dumb_tasks():
    cfg, o = _env(2, seed=4, motor_noise=0, auto_reset=0, substeps=0, observe_lag=0, hit_prob=0.0)
    b = Blob(np.zeros(o.state_words(), np.uint32), 2, 37)
    P = cfg.n_pursuers
    for e in range(2):
        for s in range(37):
            b.place(e, s, (40.0 + s, 0, 0), armed=0)
        for p in range(P):
            b.place(e, p, (1.0 + p, 0.5, 2.0)); b.set_i(e, p, "MUNITION", 455); b.set_i(e, p, "LAST_FIRED", -60)
        b.place(e, P, (1.1, 0.5, 2.0))                      # an invader 0.1 m from the agent: explosion (the agent has munition)
        b.place(e, P + 1, (0.0, 9.0, 2.0))
        b.set_ei(e, "ROUND", 3); b.set_ei(e, "MAX_STEP", 300); b.set_ei(e, "STEP", 10); b.set_ei(e, "EPISODE", 1); b.set_ef(e, "LAST_DIST", 5.0)
        b.refresh_snapshot(e)
    o.set_state(b.w)
    st, m, inert, la, act, r, d, info = o.step_students()
    a = Blob(o.get_state(), 2, 37)
    assert a.i(0, 0, "ARMED") == 0 and a.i(0, P, "ARMED") == 0 and not d.any()      # the agent exploded; the episode goes on (:600-606)
    assert act[0].tolist() == [0, 1, 1, 1, 1, 1, 1]
    # compute_reward (:452-553) by hand: gun ready (step 11, never fired) -> score = -cur; ally 1 is the agent's closest ally, its closest
    # invader is the one at (1.1, 0.5, 2) -> cur = 0.1; the agent fired and missed (hit_prob 0): reloading -> score = +cur, closeness
    # penalty (5 - 0.1) / 5 * 500; the explosion costs 1000; |p| = 2.29 < 4: no border term
    np.testing.assert_allclose(r[0], 0.1 - (4.9 / 5.0) * 500.0 - 1000.0, rtol=1e-5)
    # clipping at -3000: put the agent far outside the zone and below the floor as well
    b.set_f(0, 0, "POS", [3000.0, 0, -7.0]); b.set_f(0, 0, "OBS_POS", [3000.0, 0, -7.0])
    b.refresh_snapshot(0)
    o.set_state(b.w)
    *_, r, d, info = o.step_students()
    assert r[0] == -3000.0 and d[0] == 1
