"""The oracle's navigators in place — the ally's behaviour tree (a4) and every invader's kamikaze state machine (a3), with the stale
offsets, gun clocks and kills of a real env.step around them — against the REFERENCE's own LoyalWingmanBehaviorTree and
KamikazeNavigator run inside Exp03_vFinal_Task (tests/golden/drive_logic.npz, gen_drive_logic.py).  tests/test_gpu_fixtures.py replays
the same arenas through the C ABI on the GPU."""
import numpy as np
import pytest

from tests import _task_logic as T
from tests._blob import Blob


@pytest.fixture(scope="module")
def g(golden):
    return golden("drive_logic.npz")


def test_fixture_covers_the_branches(g):
    P = int(g["P"])
    ally1, ally2 = g["cmd1"][:, 1], g["cmd2"][:, 1]
    assert (~np.isnan(ally1[:, 0])).sum() >= 200 and ((~np.isnan(ally2[:, 0])) & (g["comparable"] == 1)).sum() >= 60
    # the tree's three leaves: chase (gun ready), formation (cooling down with munition), sacrifice = chase with no munition
    ready = (g["munition"][:, 1] == 0) | (g["step"] - 1 - g["last_fired"][:, 1] >= 60)
    assert (ready & (g["munition"][:, 1] > 0)).sum() >= 50 and (~ready).sum() >= 30 and (g["munition"][:, 1] == 0).sum() >= 30
    # all three invader states on both sides of a transition
    a = g["nav1"] >= 0
    for s0 in range(3):
        assert ((g["nav"] == s0) & a).sum() >= 100
    assert ((g["nav1"] != g["nav"]) & a).sum() >= 200
    assert g["comparable"].sum() >= 100


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_oracle_commands_what_the_reference_navigators_command(g, prec):
    from oracle import te_oracle as O
    cfg = T.config(O.default_config, g)
    orc = O.OracleEnv(cfg, prec)
    orc.set_state(T.build_blob_drive(g, orc.state_words()).w)
    n, D = cfg.n_envs, cfg.n_drones
    zeros = np.zeros((n, 4), np.float32)
    orc.step(zeros, terminal=False)
    c1, s1 = T.compare_commands(g, Blob(orc.get_state(), n, D), 1)
    orc.step(zeros, terminal=False)
    c2, s2 = T.compare_commands(g, Blob(orc.get_state(), n, D), 2)
    assert c1 >= 600 and s1 >= 400 and c2 >= 500 and s2 >= 400, (c1, s1, c2, s2)
