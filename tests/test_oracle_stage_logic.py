"""The oracle's stage01 / stage02 logic (BASELINE configs 1 and 2) against the REFERENCE's own arithmetic:
tests/golden/stage_logic.npz holds 224 stage02 arenas on which gen_stage_logic.py ran L3Stage1.on_step_middle / on_step_end with the
level3 OffsetHandler, QuadcopterManager and Gun, and 160 stage01 arenas through PyflytL2EnviromentModifiedV2.compute_reward /
compute_termination / replace_invader_if_close / update_last_distance.  tests/test_gpu_fixtures.py replays the same arenas through
the C ABI on the GPU."""
import numpy as np
import pytest

from tests import _stage_logic as S
from tests._blob import Blob


@pytest.fixture(scope="module")
def g(golden):
    return golden("stage_logic.npz")


def test_fixture_covers_the_branches(g):
    c = g["s2_counts"]   # (successful shots, pursuers exploded) as compute_reward received them
    assert len(c) >= 200 and (c[:, 0] > 0).sum() >= 50 and (c[:, 0] > 1).sum() >= 10 and (c[:, 1] > 0).sum() >= 20
    assert ((c[:, 0] > 0) & (c[:, 1] > 0)).sum() >= 5                       # shot dead and still exploding (stale matrix)
    assert 40 <= g["s2_done"].sum() <= len(c) - 40
    assert (g["s2_last_min"] - g["s2_cur_min"] > 0.01).sum() >= 30            # approach bonus candidates
    gs = g["s2_gun_state"]
    assert ((gs[:, 2] == 0) & (gs[:, 0] > 0)).sum() >= 20                     # reloading: score = d (2 reload - 1)
    assert (g["s2_munition"][:, 0] == 0).sum() >= 20                          # the suicide rule
    assert g["s2_respawned"].sum() >= 100
    assert g["s1_replaced"].sum() >= 15 and 30 <= g["s1_done"].sum() <= len(g["s1_done"]) - 30


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_oracle_reproduces_the_reference_stage02_logic(g, prec):
    from oracle import te_oracle as O
    cfg = S.config02(O.default_config, g)
    orc = O.OracleEnv(cfg, prec)
    orc.set_state(S.build_blob02(g, orc.state_words()).w)
    n = cfg.n_envs
    _, _, _, reward, done, _ = orc.step(np.zeros((n, 4), np.float32), terminal=False)
    assert S.compare02(g, reward, done, Blob(orc.get_state(), n, cfg.n_drones)) == n


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_oracle_reproduces_the_reference_stage01_logic(g, prec):
    from oracle import te_oracle as O
    cfg = S.config01(O.default_config, g)
    orc = O.OracleEnv(cfg, prec)
    orc.set_state(S.build_blob01(g, orc.state_words()).w)
    n = cfg.n_envs
    _, _, _, reward, done, _ = orc.step(np.zeros((n, 4), np.float32), terminal=False)
    assert S.compare01(g, reward, done, Blob(orc.get_state(), n, 3)) == n


def test_stage02_gun_state_matches_the_reference(g):
    """gun_state as compute_reward received it (munition / 4, reload progress, available) against the oracle's gun_state()."""
    from oracle import te_oracle as O
    gs = g["s2_gun_state"]
    for e in range(len(gs)):
        # the state AFTER this step's shot: munition_after, last_fired_after
        mun, lf, step = int(g["s2_munition_after"][e, 0]), int(g["s2_last_fired_after"][e, 0]), int(g["s2_step"][e])
        if not g["s2_armed_mid"][e, 0]:
            continue
        wait = max(60.0 - (step - lf), 0.0)
        want = [mun / 4.0, wait / 60.0, 1.0 if mun == 0 or 60.0 <= step - lf else 0.0]
        np.testing.assert_allclose(gs[e], want, atol=1e-12)
