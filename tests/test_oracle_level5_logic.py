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
    assert (c[:, 0] > 0).sum() >= 30 and (c[:, 1] > 1).sum() >= 20 and (c[:, 2] > 0).sum() >= 20 and (c[:, 3] > 0).sum() >= 20
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
