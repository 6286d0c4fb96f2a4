"""The oracle's level5 task logic with SIX wingmen — engagement loops in registry order, the reward target taken through the agent's
closest ally, five behaviour trees, the level5 round table — against the REFERENCE's own Level5_Task, EntitiesManager, OffsetHandler,
navigators and Gun run through a whole step cycle on 256 arenas (tests/golden/level5_logic.npz, gen_level5_logic.py).
tests/test_gpu_fixtures.py replays the same arenas through the C ABI on the GPU."""
import numpy as np
import pytest

from tests import _task_logic as T
from tests._blob import Blob


@pytest.fixture(scope="module")
def g(golden):
    return golden("level5_logic.npz")


def test_fixture_covers_the_branches(g):
    c = g["counts"]
    assert (c[:, 0] > 0).sum() >= 20 and (c[:, 1] > 1).sum() >= 10 and (c[:, 2] > 0).sum() >= 20 and (c[:, 3] > 0).sum() >= 12
    assert (g["closest_ally"] > 1).sum() >= 100 and (g["closest_ally"] < 0).sum() >= 10   # the target goes through allies other than slot 1, or the agent itself
    assert 40 <= g["done"].sum() <= len(g["done"]) - 100 and (g["round_after"] != g["round"]).sum() >= 10


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_oracle_reproduces_the_reference_level5_step_cycle(g, prec):
    from oracle import te_oracle as O
    cfg = T.config5(O.default_config, g)
    assert (cfg.n_pursuers, cfg.n_invaders, cfg.n_rounds, cfg.munition) == (int(g["P"]), int(g["I"]), 8, 20)
    orc = O.OracleEnv(cfg, prec)
    orc.set_state(T.build_blob_drive(g, orc.state_words()).w)
    n, D = cfg.n_envs, cfg.n_drones
    zeros = np.zeros((n, 4), np.float32)
    out = orc.step_stacked(zeros, terminal=False)
    reward, done, info = out[-3], out[-2], out[-1]
    after = Blob(orc.get_state(), n, D)
    assert T.compare(g, reward, done, info, after) == n
    c1, s1 = T.compare_commands(g, after, 1)
    orc.step_stacked(zeros, terminal=False)
    c2, s2 = T.compare_commands(g, Blob(orc.get_state(), n, D), 2)
    assert c1 >= 900 and s1 >= 250 and c2 >= 700 and s2 >= 250, (c1, s1, c2, s2)
    orc.reset()
    assert T.compare_reset(g, Blob(orc.get_state(), n, D)) == n


# ---------------------------------------------------------------------------------------------------------------- Level5DumbMultiObjectTask
@pytest.fixture(scope="module")
def gd(golden):
    return golden("level5_dumb_logic.npz")


def test_dumb_fixture_covers_the_branches(gd):
    c = gd["counts"]
    assert (c[:, 0] > 0).sum() >= 25 and (c[:, 1] > 1).sum() >= 20 and (c[:, 2] > 0).sum() >= 20 and (c[:, 3] > 0).sum() >= 20
    assert (gd["armed"][:, 0] == 0).sum() >= 10                              # the episode goes on without the agent
    assert (gd["closest_ally"] < 0).sum() >= 10 and 30 <= gd["done"].sum() <= len(gd["done"]) - 100
    assert (gd["round"] > 20).sum() >= 20                                   # rounds with 25+ invaders armed: slots beyond 32


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_oracle_reproduces_the_reference_dumb_multiobject_step_cycle(gd, prec):
    """Level5DumbMultiObjectTask: seven wingmen ALL flown by the behaviour tree, 30 invader slots (64-bit masks), its own reward, no
    termination on the agent's death: te_step_students without physics."""
    from oracle import te_oracle as O
    g = gd
    cfg = T.config5_dumb(O.default_config, g)
    assert (cfg.n_pursuers, cfg.n_invaders, cfg.n_rounds, cfg.munition) == (int(g["P"]), int(g["I"]), 26, 455)
    orc = O.OracleEnv(cfg, prec)
    orc.set_state(T.build_blob_drive(g, orc.state_words()).w)
    n, D = cfg.n_envs, cfg.n_drones
    out = orc.step_students()
    reward, done, info = out[-3], out[-2], out[-1]
    after = Blob(orc.get_state(), n, D)
    assert T.compare(g, reward, done, info, after) == n
    c1, s1 = T.compare_commands(g, after, 1)
    orc.step_students()
    c2, s2 = T.compare_commands(g, Blob(orc.get_state(), n, D), 2)
    assert c1 >= 1000 and s1 >= 250 and c2 >= 800 and s2 >= 250, (c1, s1, c2, s2)
    orc.reset()
    assert T.compare_reset(g, Blob(orc.get_state(), n, D)) == n


# ---------------------------------------------------------------------------------------------------------------- Level52BTEvaluationTask
@pytest.fixture(scope="module")
def g2(golden):
    return golden("level5_2bt_logic.npz")


def test_2bt_fixture_covers_the_branches(g2):
    g = g2
    kills = g["kills_after"][:, :2] - g["kills"][:, :2]
    assert (kills[:, 0] > 0).sum() >= 30 and (kills[:, 1] > 0).sum() >= 30 and (g["kills_after"][:, 2] > g["kills"][:, 2]).sum() >= 20
    assert (g["reward"] == 0).all() and (g["max_step_after"] == 1300).all()     # no reward, a fixed limit (a hit does not extend it)
    assert ((g["step"] > 1300) & (g["done"] == 1)).sum() >= 20 and ((g["step"] <= 1300) & (g["done"] == 0)).sum() >= 80
    origin = (np.linalg.norm(g["pos"][:, 2:], axis=2) < 0.2) & (g["armed"][:, 2:] == 1)
    assert origin.any(1).sum() >= 10                                            # the invaders-in-origin rule stays on in this task


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_oracle_reproduces_the_reference_2bt_evaluation_step_cycle(g2, prec):
    from oracle import te_oracle as O
    g = g2
    cfg = T.config5_2bt(O.default_config, g)
    assert (cfg.n_pursuers, cfg.n_invaders, cfg.n_rounds, cfg.munition, cfg.max_step, cfg.step_increment) == (2, 30, 26, 455, 1300, 0)
    orc = O.OracleEnv(cfg, prec)
    orc.set_state(T.build_blob_drive(g, orc.state_words()).w)
    n, D = cfg.n_envs, cfg.n_drones
    zeros = np.zeros((n, 4), np.float32)
    out = orc.step(zeros, terminal=False)
    reward, done, info = out[-3], out[-2], out[-1]
    after = Blob(orc.get_state(), n, D)
    assert T.compare(g, reward, done, info, after) == n
    # kills_per_drone = the wingmen's own kill words (the blob started them at 0: this step's kills)
    for e in range(n):
        assert [after.i(e, p, "KILLS") for p in range(2)] == list(g["kills_after"][e, :2] - g["kills"][e, :2]), e
    c1, s1 = T.compare_commands(g, after, 1)
    orc.step(zeros, terminal=False)
    c2, s2 = T.compare_commands(g, Blob(orc.get_state(), n, D), 2)
    assert c1 >= 800 and s1 >= 500 and c2 >= 600 and s2 >= 500, (c1, s1, c2, s2)
    orc.reset()
    assert T.compare_reset(g, Blob(orc.get_state(), n, D), per_wingman_kills=True) == n


# ---------------------------------------------------------------------------------------------------------------- Level5C1FusionTask
@pytest.fixture(scope="module")
def gc(golden):
    return golden("level5_c1_logic.npz")


def test_c1_fixture_covers_the_branches(gc):
    g = gc
    c = g["counts"]
    assert (c[:, 0] > 0).sum() >= 40 and (c[:, 4] > 0).sum() >= 15          # agent kills (+1000 each), agent suicides (-2000)
    assert (g["last_dist"] == 0).sum() >= 20                                # the first reward of an env: last_distance is born here
    d_agent = np.array([min([np.linalg.norm(g["pos"][e, 0] - g["pos"][e, 2 + j]) for j in range(10) if g["armed"][e, 2 + j]] or [np.linalg.norm(g["pos"][e, 0])])
                        for e in range(len(c))])
    closer = (d_agent < g["last_dist"]) & (g["last_dist"] > 0)
    assert closer.sum() >= 50 and (~closer).sum() >= 50                      # the 10 |v| term on and off
    assert np.abs(g["reward"]).max() <= 3000.0 + 1e-9


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_oracle_reproduces_the_reference_c1_fusion_step_cycle(gc, prec):
    from oracle import te_oracle as O
    g = gc
    cfg = T.config5_c1(O.default_config, g)
    assert (cfg.n_pursuers, cfg.n_invaders, cfg.n_rounds, cfg.munition, cfg.initial_invaders, cfg.stacked_obs) == (2, 10, 7, 49, 4, 1)
    orc = O.OracleEnv(cfg, prec)
    orc.set_state(T.build_blob_drive(g, orc.state_words()).w)
    n, D = cfg.n_envs, cfg.n_drones
    zeros = np.zeros((n, 4), np.float32)
    out = orc.step_stacked(zeros, terminal=False)
    after = Blob(orc.get_state(), n, D)
    assert T.compare(g, out[-3], out[-2], out[-1], after) == n
    c1, s1 = T.compare_commands(g, after, 1)
    orc.step_stacked(zeros, terminal=False)
    c2, s2 = T.compare_commands(g, Blob(orc.get_state(), n, D), 2)
    assert c1 >= 300 and s1 >= 250 and c2 >= 250 and s2 >= 200, (c1, s1, c2, s2)
    final = Blob(orc.get_state(), n, D)
    orc.reset()
    back = Blob(orc.get_state(), n, D)
    g_reset = {k: g[k] for k in g.files}
    g_reset["reset_last_dist"] = np.array([final.ef(e, "LAST_DIST")[0] for e in range(n)])   # the once-only last_distance: whatever the env holds stays
    assert T.compare_reset(g_reset, back) == n


def test_c1_last_distance_outlives_a_reset(gc):
    """`self.last_distance` is an attribute the task never clears (init_globals does not know it): TE_E_LAST_DIST keeps it across te_reset."""
    from oracle import te_oracle as O
    cfg = O.default_config("level5_c1", n_envs=4, motor_noise=0, seed=3)
    orc = O.OracleEnv(cfg, "f64")
    orc.reset()
    assert all(Blob(orc.get_state(), 4, cfg.n_drones).ef(e, "LAST_DIST")[0] == 0.0 for e in range(4))      # not measured yet
    orc.step_stacked(np.zeros((4, 4), np.float32), terminal=False)
    first = [float(Blob(orc.get_state(), 4, cfg.n_drones).ef(e, "LAST_DIST")[0]) for e in range(4)]
    assert all(f > 0 for f in first)
    for _ in range(3):
        orc.step_stacked(np.zeros((4, 4), np.float32), terminal=False)
    orc.reset()
    assert [float(Blob(orc.get_state(), 4, cfg.n_drones).ef(e, "LAST_DIST")[0]) for e in range(4)] == first


# ---------------------------------------------------------------------------------------------------------------- Level5FusionTask
@pytest.fixture(scope="module")
def gf(golden):
    return golden("level5_fusion_logic.npz")


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_oracle_reproduces_the_reference_fusion_step_cycle(gf, prec):
    """Level5FusionTask: the RL agent + five scripted wingmen, 30 invader slots (36 drones per env), five more invaders per round."""
    from oracle import te_oracle as O
    g = gf
    c = g["counts"]
    assert (c[:, 0] > 0).sum() >= 20 and (c[:, 1] > 1).sum() >= 15 and (g["round"] >= 5).sum() >= 40 and (g["round_after"] != g["round"]).sum() >= 3
    cfg = T.config5_fusion(O.default_config, g)
    assert (cfg.n_pursuers, cfg.n_invaders, cfg.n_rounds, cfg.munition, cfg.invaders_per_round, cfg.agent_scripted, cfg.agent_death_terminates) == (6, 30, 6, 105, 5, 0, 1)
    orc = O.OracleEnv(cfg, prec)
    orc.set_state(T.build_blob_drive(g, orc.state_words()).w)
    n, D = cfg.n_envs, cfg.n_drones
    zeros = np.zeros((n, 4), np.float32)
    out = orc.step_stacked(zeros, terminal=False)
    after = Blob(orc.get_state(), n, D)
    assert T.compare(g, out[-3], out[-2], out[-1], after) == n
    c1, s1 = T.compare_commands(g, after, 1)
    orc.step_stacked(zeros, terminal=False)
    c2, s2 = T.compare_commands(g, Blob(orc.get_state(), n, D), 2)
    assert c1 >= 900 and s1 >= 250 and c2 >= 700 and s2 >= 250, (c1, s1, c2, s2)
    orc.reset()
    assert T.compare_reset(g, Blob(orc.get_state(), n, D)) == n
