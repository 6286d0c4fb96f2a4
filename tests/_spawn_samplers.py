"""tests/golden/spawn_samplers.npz (made by tests/golden/gen_spawn_samplers.py: the reference's generate_positions of exp03 and stage02 and
stage01's U(-1, 1)^3 draws, run on recorded u = the product's own Philox words of env e, seed 0, episode 1) replayed through an engine
(the oracle, or the C ABI on the GPU): reset, wave advance, respawn.  Env e of the engine is env e of the fixture, so every position the
engine samples must be the one the REFERENCE's arithmetic makes of the same u.  Steps run with cfg.substeps = 0, cfg.observe_lag = 0."""
import numpy as np

from tests._blob import Blob

ATOL = 3e-6   # float32 positions up to 6 m from float64 reference values: polynomial sin / cos / acos on the GPU are within 1 ulp of libm


class Engine:
    """What a replay needs from an env implementation; `make(cfg)` returns the env (OracleEnv or BatchedEnv)."""

    def __init__(self, make, default_config, load, state, zeros):
        self.make, self.default_config, self.load, self.state, self.zeros = make, default_config, load, state, zeros


def _reset_into_episode(eng, env, n, D, episode):
    """te_create leaves every env freshly reset in episode 1; put the counter back and reset explicitly: the draws of te_reset are keyed on
    the episode it starts."""
    st = eng.state(env, n, D)
    for e in range(n):
        st.set_ei(e, "EPISODE", episode - 1)
    eng.load(env, st)
    env.reset()
    return eng.state(env, n, D)


def _check(got, want, what):
    np.testing.assert_allclose(np.asarray(got, np.float64), want, rtol=0, atol=ATOL, err_msg=what)


def replay_exp03(g, eng: Engine, task="exp03"):
    n, P, I = int(g["n_envs"]), 2, 9
    cfg = eng.default_config(task, n_envs=n, seed=int(g["seed"]), substeps=0, observe_lag=0, motor_noise=0, auto_reset=0)
    env = eng.make(cfg)
    D = cfg.n_drones
    st = _reset_into_episode(eng, env, n, D, int(g["episode"]))
    checked = 0
    for e in range(n):
        assert st.ei(e, "EPISODE") == int(g["episode"]) and st.ei(e, "ROUND") == 1
        for s in range(P):   # replace_pursuers / spawn_pursuer_squad: generate_positions(P, 2)
            _check(st.f(e, s, "POS", 3), g["l4_pursuer_pos"][e, s], f"env {e} pursuer {s}")
            _check(st.f(e, s, "FORMATION", 3), g["l4_pursuer_pos"][e, s], f"env {e} pursuer {s} formation")
        _check(st.f(e, P, "POS", 3), g["l4_invader_pos"][e, 0, 0], f"env {e} round 1")   # setup_round(1): generate_positions(1, 6)
        assert [st.i(e, s, "ARMED") for s in range(D)] == [1] * (P + 1) + [0] * (I - 1)
        checked += P + 1
    for rnd in range(2, I + 1):   # advance_round -> setup_round(rnd): generate_positions(rnd, 6), positions[i] -> invader i
        for e in range(n):
            for s in range(P, D):
                st.set_i(e, s, "ARMED", 0)
            st.set_ei(e, "ROUND", rnd - 1)
            st.refresh_snapshot(e)
        eng.load(env, st)
        _, _, _, _, done, _ = env.step(eng.zeros(n), terminal=False)
        assert not np.asarray(done.cpu() if hasattr(done, "cpu") else done).any()
        st = eng.state(env, n, D)
        for e in range(n):
            assert st.ei(e, "ROUND") == rnd
            for i in range(I):
                assert st.i(e, P + i, "ARMED") == (1 if i < rnd else 0), (e, rnd, i)
                if i < rnd:
                    _check(st.f(e, P + i, "POS", 3), g["l4_invader_pos"][e, rnd - 1, i], f"env {e} round {rnd} invader {i}")
                    checked += 1
    env.close()
    return checked


def replay_stage02(g, eng: Engine):
    n, P, I = int(g["n_envs"]), 2, 8
    cfg = eng.default_config("stage02", n_envs=n, n_invaders=I, seed=int(g["seed"]), substeps=0, observe_lag=0, motor_noise=0, auto_reset=0)
    env = eng.make(cfg)
    D = cfg.n_drones
    st = _reset_into_episode(eng, env, n, D, int(g["episode"]))
    checked = 0
    for e in range(n):
        for s in range(P):      # replace_pursuers: generate_positions(P, 1)
            _check(st.f(e, s, "POS", 3), g["s2_pursuer_pos"][e, s], f"env {e} pursuer {s}")
        for j in range(I):      # replace_disarmed_invaders: generate_positions(I, 2, 6)
            _check(st.f(e, P + j, "POS", 3), g["s2_invader_pos"][e, j], f"env {e} invader {j}")
        checked += D
    for k, step in enumerate(g["s2_respawn_steps"]):   # killed invaders come back inside the same step (stages.py:167-174, 371-376)
        for e in range(n):
            for s in range(P, D):
                st.set_i(e, s, "ARMED", 0)
            st.set_ei(e, "STEP", int(step) - 1)
            st.refresh_snapshot(e)
        eng.load(env, st)
        env.step(eng.zeros(n), terminal=False)
        st = eng.state(env, n, D)
        for e in range(n):
            for j in range(I):
                assert st.i(e, P + j, "ARMED") == 1
                _check(st.f(e, P + j, "POS", 3), g["s2_respawn_pos"][e, k, j], f"env {e} step {step} invader {j}")
                checked += 1
    env.close()
    return checked


def replay_stage01(g, eng: Engine):
    n = int(g["n_envs"])
    cfg = eng.default_config("stage01", n_envs=n, seed=int(g["seed"]), substeps=0, observe_lag=0, motor_noise=0, auto_reset=0)
    env = eng.make(cfg)
    st = _reset_into_episode(eng, env, n, 3, int(g["episode"]))
    checked = 0
    for e in range(n):          # reset (:101-115): invader, pursuer, extra pursuer, each U(-1, 1)^3
        for s in range(3):
            _check(st.f(e, s, "POS", 3), g["s1_pos"][e, s], f"env {e} slot {s}")
            checked += 1
    for k, step in enumerate(g["s1_catch_steps"]):   # replace_invader_if_close (:147-154)
        for e in range(n):
            p0 = st.f(e, 0, "POS", 3).astype(np.float64)
            near = p0 + np.array([0.1, -0.05, 0.02])
            st.place(e, 2, near)
            st.set_ei(e, "STEP", int(step) - 1)
            st.set_ef(e, "LAST_DIST", 5.0)
        eng.load(env, st)
        env.step(eng.zeros(n), terminal=False)
        st = eng.state(env, n, 3)
        for e in range(n):
            _check(st.f(e, 2, "POS", 3), g["s1_catch_pos"][e, k], f"env {e} step {step}")
            checked += 1
    env.close()
    return checked
