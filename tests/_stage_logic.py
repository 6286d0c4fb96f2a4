"""tests/golden/stage_logic.npz (made by tests/golden/gen_stage_logic.py from the reference's L3Stage1 / level3 OffsetHandler /
QuadcopterManager / Gun and from PyflytL2EnviromentModifiedV2's reward / termination / replace methods) as state blobs, and the
comparison of a stepped blob with what the reference produced.  Replay as in tests/_task_logic.py: cfg.substeps = 0 and
cfg.observe_lag = 0 skip the physics, arena i is env i (same hit draws)."""
import numpy as np

from tests._blob import Blob


# ------------------------------------------------------------------------------------------------------------------- stage02
def config02(default_config, g, **extra):
    n = len(g["s2_step"])
    return default_config("stage02", n_envs=n, n_invaders=int(g["s2_I"]), substeps=0, observe_lag=0, motor_noise=0, auto_reset=0,
                          seed=int(g["seed"]), dome_radius=float(g["s2_dome"]), **extra)


def build_blob02(g, words: int) -> Blob:
    n, P, I = len(g["s2_step"]), int(g["s2_P"]), int(g["s2_I"])
    D = P + I
    b = Blob(np.zeros(words, np.uint32), n, D)
    for e in range(n):
        for s in range(D):
            b.place(e, s, g["s2_pos"][e, s], armed=int(g["s2_armed"][e, s]))
            b.set_i(e, s, "MUNITION", int(g["s2_munition"][e, s]) if s < P else 10)
            b.set_i(e, s, "LAST_FIRED", int(g["s2_last_fired"][e, s]) if s < P else -60)
        b.set_f(e, 0, "VEL", g["s2_vel"][e]); b.set_f(e, 0, "OBS_VEL", g["s2_vel"][e])   # identity attitude: body = world
        b.set_ei(e, "STEP", int(g["s2_step"][e]) - 1)
        b.set_ei(e, "MAX_STEP", 600); b.set_ei(e, "EPISODE", int(g["episode"]))
        b.set_ef(e, "PREV_SNAP_MIN", g["s2_last_min"][e]); b.set_ef(e, "LAST_DIST", g["s2_last_min"][e])
        b.refresh_snapshot(e)
    return b


def compare02(g, reward, done, after: Blob):
    n, P, I = len(g["s2_step"]), int(g["s2_P"]), int(g["s2_I"])
    D = P + I
    assert np.array_equal(done.astype(bool), g["s2_done"].astype(bool)), np.flatnonzero(done.astype(bool) != g["s2_done"].astype(bool))
    np.testing.assert_allclose(reward, g["s2_reward"], rtol=3e-6, atol=2e-3)
    for e in range(n):
        armed = np.array([after.i(e, s, "ARMED") for s in range(D)])
        assert np.array_equal(armed != 0, g["s2_armed_after"][e] != 0), (e, armed, g["s2_armed_after"][e])
        for s in range(D):
            p = after.f(e, s, "POS", 3)
            if g["s2_respawned"][e, s]:     # killed this step: re-armed at a fresh draw of the reference's sampler (r in [2, 6], upper half space)
                r = float(np.linalg.norm(p))
                assert 2 - 1e-4 <= r <= 6 + 1e-4 and p[2] >= -1e-6, (e, s, p)
                assert not np.allclose(p, g["s2_pos"][e, s], atol=1e-6)
            else:
                np.testing.assert_allclose(p, g["s2_pos"][e, s], rtol=0, atol=1e-6)
        for p in range(P):
            if g["s2_armed_after"][e, p]:
                assert after.i(e, p, "MUNITION") == g["s2_munition_after"][e, p], (e, p)
                assert after.i(e, p, "LAST_FIRED") == g["s2_last_fired_after"][e, p], (e, p)
        np.testing.assert_allclose(after.ef(e, "PREV_SNAP_MIN")[0], g["s2_last_min_after"][e], rtol=2e-6, atol=1e-5)
        assert after.ei(e, "STEP") == g["s2_step"][e]
    return n


# ------------------------------------------------------------------------------------------------------------------- stage01
def config01(default_config, g, **extra):
    n = len(g["s1_step"])
    return default_config("stage01", n_envs=n, substeps=0, observe_lag=0, motor_noise=0, auto_reset=0, seed=int(g["seed"]),
                          dome_radius=float(g["s1_dome"]), **extra)


def build_blob01(g, words: int) -> Blob:
    n = len(g["s1_step"])
    b = Blob(np.zeros(words, np.uint32), n, 3)
    for e in range(n):
        for s in range(3):
            b.place(e, s, g["s1_pos"][e, s], armed=1)
            b.set_i(e, s, "MUNITION", 0 if s < 2 else 10); b.set_i(e, s, "LAST_FIRED", -60)
        b.set_f(e, 0, "VEL", g["s1_vel"][e]); b.set_f(e, 0, "OBS_VEL", g["s1_vel"][e])
        b.set_ei(e, "STEP", int(g["s1_step"][e]) - 1)           # step_calls += 1 comes first (pyflyt_level2_environment_modified_v2.py:128)
        b.set_ei(e, "MAX_STEP", 300); b.set_ei(e, "EPISODE", int(g["episode"]))
        b.set_ef(e, "LAST_DIST", g["s1_last_dist"][e])
        b.refresh_snapshot(e)
    return b


def compare01(g, reward, done, after: Blob):
    n = len(g["s1_step"])
    assert np.array_equal(done.astype(bool), g["s1_done"].astype(bool)), np.flatnonzero(done.astype(bool) != g["s1_done"].astype(bool))
    np.testing.assert_allclose(reward, g["s1_reward"], rtol=3e-6, atol=2e-3)
    for e in range(n):
        inv, p0 = after.f(e, 2, "POS", 3), after.f(e, 0, "POS", 3)
        if g["s1_replaced"][e]:            # caught: the invader restarts at a fresh U(-1, 1)^3 draw
            assert np.all(np.abs(inv) <= 1 + 1e-6) and not np.allclose(inv, g["s1_pos"][e, 2], atol=1e-6), (e, inv)
            np.testing.assert_allclose(after.ef(e, "LAST_DIST")[0], np.linalg.norm(inv.astype(np.float64) - p0), rtol=1e-5, atol=1e-5)
        else:
            np.testing.assert_allclose(inv, g["s1_pos"][e, 2], rtol=0, atol=1e-6)
            np.testing.assert_allclose(after.ef(e, "LAST_DIST")[0], g["s1_last_dist_after"][e], rtol=2e-6, atol=1e-5)
        assert after.ei(e, "STEP") == g["s1_step"][e]
    return n
