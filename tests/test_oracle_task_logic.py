"""The oracle's task logic (offsets, engagement order, reward, termination, info, wave advance) against the REFERENCE's own
arithmetic: tests/golden/task_logic.npz holds 288 arenas on which gen_task_logic.py ran the reference's OffsetHandler,
EntitiesManager, Gun and Exp03_vFinal_Task.on_step_middle / on_step_end.  The same arenas are replayed through the C ABI on the
GPU by tests/test_gpu_fixtures.py."""
import numpy as np
import pytest

from tests import _task_logic as T
from tests._blob import Blob


@pytest.fixture(scope="module")
def g(golden):
    return golden("task_logic.npz")


def test_fixture_covers_the_branches(g):
    c = g["counts"]  # (agent shots, ally shots, exploded, ally suicides, agent suicides) as compute_reward received them
    assert len(g["step"]) >= 200
    assert (c[:, 0] > 0).sum() >= 20 and (c[:, 1] > 0).sum() >= 20 and (c[:, 2] > 0).sum() >= 20
    assert (c[:, 3] > 0).sum() >= 5 and (c[:, 4] > 0).sum() >= 5
    assert ((c[:, 0] > 0) & (c[:, 1] > 0)).sum() >= 5                      # both pursuers credited in one step
    assert (g["round_after"] != g["round"]).sum() >= 10                     # wave advance
    assert 30 <= g["done"].sum() <= len(g["done"]) - 30
    # the C8 zone term as a bonus (4 < |p| < 8) and as a penalty (|p| > 8)
    r = np.linalg.norm(g["pos"][:, 0], axis=1)
    assert ((r > 4) & (r < 8)).sum() >= 10 and (r > 8).sum() >= 5
    # stale-matrix explosion: an invader shot dead in this step still explodes on the pursuer next to it
    shot_and_exploded = [(e) for e in range(len(c)) if c[e, 0] + c[e, 1] > 0 and c[e, 2] + c[e, 3] + c[e, 4] > 0]
    assert len(shot_and_exploded) >= 5


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_oracle_reproduces_the_reference_task_logic(g, prec):
    from oracle import te_oracle as O
    cfg = T.config(O.default_config, g)
    orc = O.OracleEnv(cfg, prec)
    orc.set_state(T.build_blob(g, orc.state_words()).w)
    n = cfg.n_envs
    _, _, _, reward, done, info = orc.step(np.zeros((n, 4), np.float32), terminal=False)
    after = Blob(orc.get_state(), n, cfg.n_drones)
    assert T.compare(g, reward, done, info, after) == n


def test_oracle_offset_queries_match_the_reference(g):
    """OffsetHandler's distance matrix and identify_* answers, recomputed from the arena positions the way the oracle's
    closest_* helpers do (strict '<' in slot order = the reference's stable sort / argmin)."""
    P, I = int(g["P"]), int(g["I"])
    for e in range(len(g["step"])):
        armed, pos = g["armed"][e], g["pos"][e]
        for p in range(P):
            if not armed[p]:
                continue
            d = np.array([np.linalg.norm(pos[p] - pos[P + j]) if armed[P + j] else np.inf for j in range(I)])
            for j in range(I):
                if armed[P + j]:
                    assert abs(d[j] - g["dist"][e, p, j]) < 1e-12
            order = [P + j for j in np.argsort(d, kind="stable") if np.isfinite(d[j])]
            assert g["closest_invader"][e, p] == order[0]
            for rng, key in ((1.0, "in_shoot"), (0.2, "in_explode")):
                want = [s for s in order if d[s - P] < rng]
                got = [s for s in g[key][e, p] if s >= 0]
                assert got == want, (e, p, key)
