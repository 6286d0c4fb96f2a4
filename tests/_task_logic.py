"""tests/golden/task_logic.npz (made by tests/golden/gen_task_logic.py from the reference's OffsetHandler, EntitiesManager, Gun and
Exp03_vFinal_Task) as a state blob, and the comparison of a stepped blob with what the reference produced.

Every arena is one environment at the moment the reference calls `task.on_step_middle()`.  Replay: cfg.substeps = 0 and
cfg.observe_lag = 0 make env.step() skip the physics (the IMU read after zero sub-steps is the loaded state: identity
attitude, POS == OBS_POS), so the step's engagement / reward / termination / wave logic runs on exactly the fixture's positions.
Arena i is env i: the hit draws the reference consumed are the product's Philox words for (seed 0, env i, episode 1, step)."""
import numpy as np

from dronechase_amd import config as K
from tests._blob import Blob


def config(default_config, g, **extra):
    n = len(g["step"])
    return default_config("exp03", n_envs=n, substeps=0, observe_lag=0, motor_noise=0, auto_reset=0, seed=int(g["seed"]),
                          dome_radius=float(g["dome"]), **extra)


def build_blob(g, words: int) -> Blob:
    n, P, I = len(g["step"]), int(g["P"]), int(g["I"])
    D = P + I
    b = Blob(np.zeros(words, np.uint32), n, D)
    for e in range(n):
        for s in range(D):
            b.place(e, s, g["pos"][e, s], armed=int(g["armed"][e, s]))
            if s < P:
                b.set_i(e, s, "MUNITION", int(g["munition"][e, s])); b.set_i(e, s, "LAST_FIRED", int(g["last_fired"][e, s]))
            else:
                b.set_i(e, s, "MUNITION", 10); b.set_i(e, s, "LAST_FIRED", -60)
        b.set_f(e, 0, "VEL", g["vel"][e]); b.set_f(e, 0, "OBS_VEL", g["vel"][e])   # identity attitude: body = world
        b.set_ei(e, "STEP", int(g["step"][e]) - 1)                                 # the step broadcast comes after the sim loop
        b.set_ei(e, "MAX_STEP", int(g["max_step"][e])); b.set_ei(e, "ROUND", int(g["round"][e]))
        b.set_ef(e, "LAST_DIST", g["last_dist"][e])
        b.set_ei(e, "AGENT_KILLS", int(g["kills"][e, 0])); b.set_ei(e, "ALLIES_KILLS", int(g["kills"][e, 1])); b.set_ei(e, "DEADS", int(g["kills"][e, 2]))
        b.set_ei(e, "EPISODE", int(g["episode"]))
        b.refresh_snapshot(e)
    return b


def compare(g, reward, done, info, after: Blob, reward_atol=2e-3):
    """reward / done / info of the step and the state it left, against the reference's numbers.  Returns the number of arenas checked."""
    n, P, I = len(g["step"]), int(g["P"]), int(g["I"])
    D = P + I
    done_ref = g["done"].astype(bool)
    assert np.array_equal(done.astype(bool), done_ref), np.flatnonzero(done.astype(bool) != done_ref)
    np.testing.assert_allclose(reward, g["reward"], rtol=2e-6, atol=reward_atol)
    assert np.array_equal(info, g["info"]), np.flatnonzero((info != g["info"]).any(1))
    for e in range(n):
        armed = np.array([after.i(e, s, "ARMED") for s in range(D)])
        # a terminal step: the reference's on_step_end is unobservable (SB3 resets the env), the product skips it
        want = g["armed_mid"][e] if done_ref[e] else g["armed_after"][e]
        assert np.array_equal(armed != 0, want != 0), (e, armed, want)
        assert [after.i(e, p, "MUNITION") for p in range(P)] == list(g["munition_after"][e]) or not done_ref[e] and g["round_after"][e] != g["round"][e], e
        fired = g["shots_fired"][e].astype(bool)
        for p in range(P):
            if fired[p]:
                assert after.i(e, p, "LAST_FIRED") == g["step"][e] and after.i(e, p, "MUNITION") == g["munition"][e, p] - 1, (e, p)
            else:
                assert after.i(e, p, "LAST_FIRED") == g["last_fired"][e, p] and after.i(e, p, "MUNITION") == g["munition"][e, p], (e, p)
        assert after.ei(e, "MAX_STEP") == g["max_step_after"][e], e
        assert [after.ei(e, "AGENT_KILLS"), after.ei(e, "ALLIES_KILLS"), after.ei(e, "DEADS")] == list(g["kills_after"][e]), e
        np.testing.assert_allclose(after.ef(e, "LAST_DIST")[0], g["last_dist_after"][e], rtol=2e-6, atol=1e-5)
        if not done_ref[e]:
            assert after.ei(e, "ROUND") == g["round_after"][e], e
        assert after.ei(e, "STEP") == g["step"][e], e
    return n


# ---------------------------------------------------------------------------------------------------------------------------------------
# tests/golden/drive_logic.npz (gen_drive_logic.py): the reference's LoyalWingmanBehaviorTree and KamikazeNavigator run inside
# Exp03_vFinal_Task around one env.step: the commands of step t (update #1) and of step t+1 (update #2), the invaders' states after each.
def build_blob_drive(g, words: int) -> Blob:
    b = build_blob(g, words)
    n, P, I = len(g["step"]), int(g["P"]), int(g["I"])
    for e in range(n):
        for j in range(I):
            b.set_i(e, P + j, "NAV_STATE", int(g["nav"][e, j]))
        if g["formation"].ndim == 3:                 # level5_logic.npz: every wingman's formation point
            for p in range(P):
                b.set_f(e, p, "FORMATION", g["formation"][e, p])
        else:
            b.set_f(e, 1, "FORMATION", g["formation"][e])
    return b


def config5(default_config, g, **extra):
    n = len(g["step"])
    return default_config("level5", n_envs=n, substeps=0, observe_lag=0, motor_noise=0, auto_reset=0, seed=int(g["seed"]),
                          dome_radius=float(g["dome"]), **extra)


def setpoint_of(cmd):
    """Quadcopter.convert_command_to_setpoint (quadcopter.py:379-396; pinned by tests/golden/command.npz): unit(direction) * magnitude as [vx, vy, 0, vz]"""
    d = np.asarray(cmd[:3], np.float64)
    n = np.linalg.norm(d)
    v = cmd[3] * d / (n if n > 0 else 1.0)
    return np.array([v[0], v[1], 0.0, v[2]])


def config5_dumb(default_config, g, **extra):
    n = len(g["step"])
    return default_config("level5_dumb", n_envs=n, substeps=0, observe_lag=0, motor_noise=0, auto_reset=0, seed=int(g["seed"]),
                          dome_radius=float(g["dome"]), **extra)


def config5_c1(default_config, g, **extra):
    n = len(g["step"])
    return default_config("level5_c1", n_envs=n, substeps=0, observe_lag=0, motor_noise=0, auto_reset=0, seed=int(g["seed"]),
                          dome_radius=float(g["dome"]), **extra)


def config5_fusion(default_config, g, **extra):
    n = len(g["step"])
    return default_config("level5_fusion", n_envs=n, substeps=0, observe_lag=0, motor_noise=0, auto_reset=0, seed=int(g["seed"]),
                          dome_radius=float(g["dome"]), **extra)


def config5_2bt(default_config, g, **extra):
    n = len(g["step"])
    return default_config("level5_2bt", n_envs=n, substeps=0, observe_lag=0, motor_noise=0, auto_reset=0, seed=int(g["seed"]),
                          dome_radius=float(g["dome"]), **extra)


def compare_commands(g, after: Blob, which: int, still_armed=None, atol=2e-6):
    """Set-point words and invader states after product step `which` (1: the commands of step t, 2: of step t+1) against the reference's
    navigators.  Only drones that are still armed when the state is read can be compared (disarm clears the set-point), and update #2 only
    in arenas where the step neither ended the episode nor started a new round.  Returns (commands compared, states compared)."""
    n, P, I = len(g["step"]), int(g["P"]), int(g["I"])
    D = P + I
    cmd, nav = g["cmd%d" % which], g["nav%d" % which]
    n_cmd = n_nav = 0
    for e in range(n):
        if which == 2 and not g["comparable"][e]:
            continue
        for s in range(0 if "agent_scripted" in g and int(g["agent_scripted"]) else 1, D):   # slot 0 is the RL agent (its set-point is the action) unless the task flies it too
            if np.isnan(cmd[e, s, 0]) or not after.i(e, s, "ARMED"):
                continue
            if which == 1 and (not g["armed_after"][e, s] or (s >= P and not g["comparable"][e])):   # a new round re-arms invaders at fresh positions
                continue
            np.testing.assert_allclose(after.f(e, s, "SETPOINT", 4), setpoint_of(cmd[e, s]), rtol=0, atol=atol, err_msg=f"arena {e} slot {s} update {which}")
            n_cmd += 1
            if s >= P:
                assert after.i(e, s, "NAV_STATE") == nav[e, s - P], (e, s, which)
                n_nav += 1
    return n_cmd, n_nav


# ---------------------------------------------------------------------------------------------------------------------------------------
# tests/golden/evaluation_logic.npz (gen_evaluation_logic.py): Evaluation_Task with two behaviour-tree drivers through a whole step cycle.
# TIME_IS_LIMITED is a property of the task object: the arenas split into two te_envs (cfg.max_step = 0: no limit).
def evaluation_groups(g):
    lim = g["limited"].astype(bool)
    return [(np.flatnonzero(lim), True), (np.flatnonzero(~lim), False)]


def evaluation_config(default_config, g, idx, limited):
    assert np.array_equal(idx, np.arange(idx[0], idx[0] + len(idx)))     # contiguous: arena i is GLOBAL env i (its hit draws are keyed on it)
    return default_config("evaluation", n_envs=len(idx), env_index_base=int(idx[0]), n_pursuers=int(g["P"]), n_invaders=int(g["I"]), n_rounds=int(g["I"]), munition=20,
                          max_step=300 if limited else 0, substeps=0, observe_lag=0, motor_noise=0, auto_reset=0, seed=int(g["seed"]), dome_radius=float(g["dome"]))


def evaluation_blob(g, idx, words: int) -> Blob:
    sub = {k: (g[k][idx] if getattr(g[k], "ndim", 0) >= 1 and len(g[k]) == len(g["step"]) else g[k]) for k in g.files}
    sub["last_dist"] = np.full(len(idx), 5.0); sub["kills"] = np.zeros((len(idx), 3), np.int32)
    b = build_blob(sub, words)
    n, P, I = len(idx), int(g["P"]), int(g["I"])
    for e in range(n):
        for j in range(I):
            b.set_i(e, P + j, "NAV_STATE", int(sub["nav"][e, j]))
        for p in range(P):
            b.set_f(e, p, "FORMATION", sub["formation"][e, p])
            b.set_i(e, p, "KILLS", int(g["kills"][idx[e], p]))          # lw_kills of the episode so far
    return b, sub


def compare_evaluation(sub, reward, done, rows, after: Blob):
    """reward 0, termination, who is armed, guns, MAX_STEP, round and the info rows (lw_kills, lw_alive, lw_munitions, current_wave, step) of the
    wingmen the reference lists (the armed ones)."""
    n, P, I = len(sub["step"]), int(sub["P"]), int(sub["I"])
    D = P + I
    assert not np.asarray(reward).any()
    assert np.array_equal(np.asarray(done).astype(bool), sub["done"].astype(bool)), np.flatnonzero(np.asarray(done).astype(bool) != sub["done"].astype(bool))
    for e in range(n):
        term = bool(sub["done"][e])
        armed = np.array([after.i(e, s, "ARMED") for s in range(D)])
        want = sub["armed_mid"][e] if term else sub["armed_after"][e]
        assert np.array_equal(armed != 0, want != 0), (e, armed, want)
        for p in range(P):
            if sub["armed_mid"][e, p] and (term or sub["round_after"][e] == sub["round"][e]):
                assert after.i(e, p, "MUNITION") == sub["munition_after"][e, p] and after.i(e, p, "LAST_FIRED") == sub["last_fired_after"][e, p], (e, p)
            assert after.i(e, p, "KILLS") == sub["lw_kills_after"][e, p], (e, p)
            if sub["info_rows"][e, p, 0] >= 0:
                assert list(rows[e, p]) == list(sub["info_rows"][e, p]), (e, p, rows[e, p], sub["info_rows"][e, p])
            else:
                assert rows[e, p, 1] == 0, (e, p)        # not listed by the reference = not alive
        assert after.ei(e, "MAX_STEP") == sub["max_step_after"][e], e
        if not term:
            assert after.ei(e, "ROUND") == sub["round_after"][e], e
    return n


def compare_reset(g, after: Blob, per_wingman_kills=False):
    """Env.reset -> Task.on_reset of the fixtures made by gen_level5_logic.py: who is armed (every wingman, the first round's invaders), guns
    (full munition, cooldown over), MAX_STEP, round 1, the counters, last_closest_distance (Level5C1FusionTask's last_distance survives), the
    invaders' state machines back in WaitState, step 0."""
    n, P, I = len(g["step"]), int(g["P"]), int(g["I"])
    for e in range(n):
        assert [int(after.i(e, s, "ARMED") != 0) for s in range(P + I)] == list(g["reset_armed"][e]), e
        assert [after.i(e, p, "MUNITION") for p in range(P)] == list(g["reset_munition"][e]) and [after.i(e, p, "LAST_FIRED") for p in range(P)] == list(g["reset_last_fired"][e]), e
        assert after.ei(e, "MAX_STEP") == g["reset_max_step"][e] and after.ei(e, "ROUND") == g["reset_round"][e] and after.ei(e, "STEP") == 0, e
        if per_wingman_kills:
            assert [after.i(e, p, "KILLS") for p in range(P)] == list(g["reset_kills"][e, :P]) and after.ei(e, "DEADS") == g["reset_kills"][e, 2], e
        else:
            assert [after.ei(e, "AGENT_KILLS"), after.ei(e, "ALLIES_KILLS"), after.ei(e, "DEADS")] == list(g["reset_kills"][e]), e
        np.testing.assert_allclose(after.ef(e, "LAST_DIST")[0], g["reset_last_dist"][e], rtol=2e-6, atol=1e-5)
        assert all(after.i(e, P + j, "NAV_STATE") == g["reset_nav"][e, j] for j in range(I) if g["reset_armed"][e, P + j]), e
    return n
