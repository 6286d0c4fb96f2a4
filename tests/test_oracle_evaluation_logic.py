"""The oracle's Evaluation_Task rules (cfg.evaluation: every pursuer obeys the behaviour tree, reward 0, no invaders-in-origin rule, optional
time limit, kills per wingman, the info rows of the armed wingmen) against the REFERENCE's own Evaluation_Task run with two "bt" drivers through
a whole step cycle on 256 arenas (tests/golden/evaluation_logic.npz, gen_evaluation_logic.py).  GPU: tests/test_gpu_fixtures.py."""
import numpy as np
import pytest

from tests import _task_logic as T
from tests._blob import Blob


@pytest.fixture(scope="module")
def g(golden):
    return golden("evaluation_logic.npz")


def test_fixture_covers_the_branches(g):
    dk = g["lw_kills_after"] - g["kills"][:, :2]
    assert (dk[:, 0] > 0).sum() >= 30 and (dk[:, 1] > 0).sum() >= 30
    lim = g["limited"].astype(bool)
    late = g["step"] > g["max_step"]
    assert (late & lim & (g["done"] == 1)).sum() >= 5 and (late & ~lim & (g["done"] == 0)).sum() >= 10     # the limit binds only under TIME_IS_LIMITED
    assert (g["armed"][:, 0] == 0).sum() >= 10 and (g["reward"] == 0).all()
    origin = (np.linalg.norm(g["pos"][:, 2:], axis=2) < 0.2) & (g["armed"][:, 2:] == 1)
    assert (origin & (g["armed_mid"][:, 2:] == 1)).any(1).sum() >= 5       # an invader in the origin stays armed: the rule is commented out (:397)


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_oracle_reproduces_the_reference_evaluation_step_cycle(g, prec):
    from oracle import te_oracle as O
    total = c1 = s1 = c2 = s2 = 0
    for idx, limited in T.evaluation_groups(g):
        cfg = T.evaluation_config(O.default_config, g, idx, limited)
        orc = O.OracleEnv(cfg, prec)
        blob, sub = T.evaluation_blob(g, idx, orc.state_words())
        orc.set_state(blob.w)
        n, D = cfg.n_envs, cfg.n_drones
        zeros = np.zeros((n, 4), np.float32)
        out = orc.step(zeros, terminal=False)
        after = Blob(orc.get_state(), n, D)
        total += T.compare_evaluation(sub, out[-3], out[-2], orc.wingman_info(), after)
        a, b = T.compare_commands(sub, after, 1); c1 += a; s1 += b
        orc.step(zeros, terminal=False)
        a, b = T.compare_commands(sub, Blob(orc.get_state(), n, D), 2); c2 += a; s2 += b
    assert total == len(g["step"]) and c1 >= 700 and s1 >= 400 and c2 >= 500 and s2 >= 350, (total, c1, s1, c2, s2)
