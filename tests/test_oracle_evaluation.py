"""Evaluation_Task rules (per-wingman info rows; level4/evaluation_environment.py:170-187,
tasks/evaluation_task.py:89-112,257-275,381-404,498-574) as the config switch cfg.evaluation: behaviour-tree drivers, and
caller-driven ones (the reference's "nn" drivers) through the driver mask.
CPU scenario tests of the oracle, each tied to the reference lines it restates; the GPU path is held to the oracle by
tests/test_gpu_evaluation.py."""
import numpy as np
import pytest

from dronechase_amd import config as K
from oracle import te_oracle as O
from tests._blob import Blob
from tests.test_oracle_tasks import arena, load, step


def make(n_pursuers=1, **over):
    over.setdefault("motor_noise", 0)
    if n_pursuers != 1:
        rounds = O.lib("f64").te_calculate_rounds(n_pursuers, 20)
        over.update(n_pursuers=n_pursuers, n_rounds=rounds, n_invaders=rounds)
    cfg = O.default_config("evaluation", n_envs=1, **over)
    env = O.OracleEnv(cfg, "f64")
    env.reset()
    return cfg, env


def test_preset_is_one_behaviour_tree_driver_without_time_limit():
    """evaluation_exp01_1bt_app_ready.py:64-68 (one "bt" driver); evaluation_task.py:89-112 defaults."""
    c = O.default_config("evaluation")
    assert (c.n_pursuers, c.munition, c.evaluation, c.ally_policy, c.max_step, c.step_increment) == (1, 20, 1, K.ALLY_BT, 0, 100)
    assert c.n_rounds == c.n_invaders == O.lib("f64").te_calculate_rounds(1, 20) == 6   # n(n+1)/2 >= 20
    assert O.lib("f64").te_calculate_rounds(2, 20) == O.default_config("exp03").n_rounds == 9


def test_pursuer_zero_obeys_the_behaviour_tree_and_ignores_the_action():
    """EvaluationEnvironment.step(actions_not_used) (evaluation_environment.py:170-187); drive_lw flies every armed
    pursuer with its driver (evaluation_task.py:257-275): gun ready -> chase the closest invader at 0.6 m/s."""
    cfg, env = make()
    arena(cfg, env, agent=(0, 0, 3), invaders=((4, 0, 3),))
    out = step(env, [0, 1, 0, 1])              # "fly along +y at full speed": must be ignored
    b = load(env, cfg)
    sp = b.f(0, 0, "SETPOINT", 4)
    np.testing.assert_allclose(sp, [cfg.ally_speed, 0, 0, 0], atol=1e-6)   # toward the invader on +x
    assert b.f(0, 0, "POS", 3)[0] > 0 and abs(b.f(0, 0, "POS", 3)[1]) < 1e-3
    assert out["reward"] == 0.0 and not out["done"]                          # compute_reward is 0 (:508-515)


def test_kills_are_counted_per_wingman_and_survive_the_waves():
    """lw_kills[pursuer.id] += 1 on a successful shot (evaluation_task.py:498-499); compute_info rows (:553-574)."""
    cfg, env = make(n_pursuers=2, hit_prob=1.0)
    arena(cfg, env, agent=(0, 0, 3), ally=(5, 5, 3), invaders=((5.5, 5, 3),))   # only the ally is in shoot range
    out = step(env)
    rows = env.wingman_info()[0]
    assert rows[:, 0].tolist() == [0, 1] and rows[:, 1].tolist() == [1, 1] and rows[:, 2].tolist() == [20, 19]
    assert list(out["info"]) == [0, 1, 0, 1]                     # the classic info still says "an ally killed"
    b = load(env, cfg)
    # the wave advanced in on_step_end, AFTER compute_info ran (evaluation_environment.py:178-186): the info rows still say wave 1
    # (pinned by tests/golden/evaluation_logic.npz); kills kept
    assert b.ei(0, "ROUND") == 2 and rows[0, 3] == 1 and rows[0, 4] == 1
    assert b.i(0, 1, "KILLS") == 1 and b.i(0, 0, "KILLS") == 0
    # a second kill by the agent in the next wave: place one invader next to it, park the other far away but inside
    b.place(0, 2, (0.5, 0, 3)); b.hover_ready(0, 2, cfg)
    b.place(0, 3, (0, -9, 3)); b.hover_ready(0, 3, cfg)
    b.place(0, 0, (0, 0, 3)); b.refresh_snapshot(0)
    env.set_state(b.w)
    step(env)
    rows = env.wingman_info()[0]
    assert rows[:, 0].tolist() == [1, 1] and rows[0, 2] == 19
    env.reset()
    assert env.wingman_info()[0][:, 0].tolist() == [0, 0]        # a new episode starts from zero


def test_termination_rules():
    """evaluation_task.py:519-551: no time limit unless TIME_IS_LIMITED, all rounds over, anybody outside the dome, all
    pursuers destroyed -- but NOT the death of pursuer 0 alone, and no invaders-in-origin rule (:397)."""
    cfg, env = make(n_pursuers=2)
    # (a) pursuer 0 explodes, the other lives on: exp03 would end here
    arena(cfg, env, agent=(0, 0, 3), ally=(8, 8, 3), invaders=((0.1, 0, 3), (0, -9, 3)))
    b = load(env, cfg); b.set_i(0, 0, "MUNITION", 0); env.set_state(b.w)
    out = step(env)
    rows = env.wingman_info()[0]
    assert not out["done"] and rows[:, 1].tolist() == [0, 1]
    # (b) an invader inside the origin range is NOT removed, and the step counter may pass 300
    cfg, env = make(n_pursuers=1)
    arena(cfg, env, agent=(8, 8, 3), invaders=((0.05, 0, 0.05),))
    b = load(env, cfg); b.set_ei(0, "STEP", 400); env.set_state(b.w)
    out = step(env)
    assert not out["done"] and load(env, cfg).i(0, cfg.n_pursuers, "ARMED") == 1
    # (c) TIME_IS_LIMITED: max_step > 0 brings the limit back
    cfg, env = make(n_pursuers=1, max_step=300)
    arena(cfg, env, agent=(8, 8, 3), invaders=((0, -9, 3),))
    b = load(env, cfg); b.set_ei(0, "STEP", 300); b.set_ei(0, "MAX_STEP", 300); env.set_state(b.w)
    assert step(env)["done"]
    # (d) an invader outside the dome ends the episode; so does the last pursuer's death
    cfg, env = make(n_pursuers=1)
    arena(cfg, env, agent=(0, 0, 3), invaders=((0, 0, 20.5),))
    assert step(env)["done"]
    cfg, env = make(n_pursuers=1, auto_reset=0)
    arena(cfg, env, agent=(0, 0, 3), invaders=((0.1, 0, 3),))
    b = load(env, cfg); b.set_i(0, 0, "MUNITION", 0); env.set_state(b.w)
    out = step(env)
    assert out["done"] and env.wingman_info()[0][0, 1] == 0


def test_free_running_episodes_end_and_rows_are_consistent():
    N = 64
    cfg = O.default_config("evaluation", n_envs=N, seed=6, motor_noise=1)   # auto-reset on: rows of a done env are the fresh episode's
    env = O.OracleEnv(cfg, "f32", threads=4)
    env.reset()
    dones = max_kills = max_wave = 0
    for t in range(600):
        _, _, _, r, d, info = env.step(np.zeros((N, 4), np.float32))
        rows = env.wingman_info()
        live = d == 0
        assert (r == 0).all()
        assert (rows[live, 0, 0] == info[live, 0]).all()                      # pursuer 0's row = the classic agent_kills
        assert (rows[live, 0, 3] >= info[live, 3]).all()                      # info holds the wave before on_step_end
        assert ((rows[..., 2] >= 0) & (rows[..., 2] <= 20)).all()
        assert (rows[..., 0] <= 20 - rows[..., 2]).all()                      # a kill costs a round
        assert (rows[~live][..., 0] == 0).all() and (rows[~live][..., 4] == 0).all()   # auto-reset: fresh episode
        dones += int(d.sum()); max_kills = max(max_kills, int(rows[..., 0].max())); max_wave = max(max_wave, int(rows[..., 3].max()))
    assert dones > 0 and max_kills >= 2 and max_wave >= 3   # the behaviour tree kills, clears waves, and episodes end


def _action_of_setpoint(sp):
    v = np.array([sp[0], sp[1], sp[3]], np.float64)
    n = np.linalg.norm(v)
    return np.array([*(v / n if n > 0 else v), n], np.float32)


def test_caller_driven_wingmen_that_answer_like_the_behaviour_tree_reproduce_the_scripted_run():
    """Evaluation_Task.drive_lw treats a driver with `predict` like exp05's ally (evaluation_task.py:257-268): observation,
    predict, drive.  cfg.evaluation's driver mask hands those pursuers to the caller (ote_observe_wingman /
    ote_set_wingman_actions); feeding back the command the behaviour tree gives in the all-scripted twin must give the same run."""
    N, T, P = 32, 120, 2
    rounds = O.lib("f64").te_calculate_rounds(P, 20)
    base = dict(n_envs=N, seed=4, motor_noise=1, n_pursuers=P, n_rounds=rounds, n_invaders=rounds, max_step=60)
    cs = O.default_config("evaluation", **base)                                   # both wingmen scripted
    cx = O.default_config("evaluation", evaluation=1 | (0b11 << 8), **base)       # both flown by the caller
    es, ex = O.OracleEnv(cs, "f64"), O.OracleEnv(cx, "f64")
    es.reset(); ex.reset()
    with pytest.raises(AssertionError):
        es.observe_wingman(0)                     # not caller-driven in the scripted twin
    zeros = np.zeros((N, 4), np.float32)
    dones = 0
    for t in range(T):
        before = Blob(es.get_state(), N, cs.n_drones)
        _, _, _, r, d, info = (x.copy() for x in es.step(zeros))
        after = Blob(es.get_state(), N, cs.n_drones)
        probe = None
        for p in range(P):
            lid, inert, last, active = ex.observe_wingman(p)
            assert (active == np.array([before.i(e, p, "ARMED") for e in range(N)])).all()
            acts = np.zeros((N, 4), np.float32)
            for e in range(N):
                if not before.i(e, p, "ARMED"):
                    continue
                if d[e]:          # auto-reset wiped the set-point: replay the step without auto-reset to read it
                    if probe is None:
                        probe = O.OracleEnv(O.default_config("evaluation", auto_reset=0, **base), "f64")
                        probe.set_state(before.w); probe.step(zeros)
                        probe = Blob(probe.get_state(), N, cs.n_drones)
                    acts[e] = _action_of_setpoint(probe.f(e, p, "SETPOINT", 4))
                else:
                    acts[e] = _action_of_setpoint(after.f(e, p, "SETPOINT", 4))
            ex.set_wingman_actions(p, acts)
        _, _, _, rx, dx, infox = ex.step(zeros)
        np.testing.assert_array_equal(dx, d); np.testing.assert_array_equal(infox, info)
        np.testing.assert_array_equal(ex.wingman_info(), es.wingman_info())
        dones += int(d.sum())
    assert dones >= 5      # resets happened (every kill extends the time limit by 100 steps: episodes are long)
    bs, bx = Blob(es.get_state(), N, cs.n_drones), Blob(ex.get_state(), N, cx.n_drones)
    keep = [i for i in range(K.DRONE_WORDS) if not (K.D["ALLY_ACTION"] <= i < K.D["ALLY_ACTION"] + 4) and i not in K.D_INT_WORDS]
    np.testing.assert_allclose(bx.dr[..., keep].view(np.float32), bs.dr[..., keep].view(np.float32), atol=1e-5)
