"""the spawn samplers against the REFERENCE's own arithmetic.  tests/golden/spawn_samplers.npz holds what
Exp03_vFinal_Task.generate_positions (exp03_vFinal_task.py:584-608), L3Stage1.generate_positions (level3/components/stages.py:350-368) and
stage01's np.random.uniform(-1, 1, 3) draws (pyflyt_level2_environment_modified_v2.py:101-115,154) return when numpy's uniform is fed the
product's own Philox words; here the oracle's reset / wave advance / respawn must land every drone on those positions.  The same bodies
run through the C ABI on the GPU in tests/test_gpu_fixtures.py."""
import numpy as np
import pytest

from tests import _spawn_samplers as SP
from tests._blob import Blob


@pytest.fixture(scope="module")
def g(golden):
    return golden("spawn_samplers.npz")


def _engine(prec):
    from oracle import te_oracle as O
    return SP.Engine(make=lambda cfg: O.OracleEnv(cfg, prec), default_config=O.default_config,
                     load=lambda env, blob: env.set_state(blob.w), state=lambda env, n, D: Blob(env.get_state(), n, D),
                     zeros=lambda n: np.zeros((n, 4), np.float32))


def test_fixture_shape_and_ranges(g):
    n = int(g["n_envs"])
    assert n >= 64 and g["l4_invader_pos"].shape == (n, 9, 9, 3) and g["s2_respawn_pos"].shape[1:] == (len(g["s2_respawn_steps"]), 8, 3)
    r = np.linalg.norm(g["l4_pursuer_pos"], axis=-1)
    np.testing.assert_allclose(r, 2.0, atol=1e-12)
    for rnd in range(1, 10):
        p = g["l4_invader_pos"][:, rnd - 1, :rnd]
        np.testing.assert_allclose(np.linalg.norm(p, axis=-1), 6.0, atol=1e-12)
        assert (p[..., 2] >= 0).all() and (p[..., 2] <= 4.0 + 1e-12).all() and (p[..., 1] >= -1e-12).all()   # z = r cos(phi), phi in [acos(4/6), pi/2]; theta in [0, pi]
        assert np.isnan(g["l4_invader_pos"][:, rnd - 1, rnd:]).all()
    r2 = np.linalg.norm(g["s2_invader_pos"], axis=-1)
    assert (r2 >= 2 - 1e-12).all() and (r2 <= 6 + 1e-12).all() and (g["s2_invader_pos"][..., 2] >= 0).all()
    np.testing.assert_allclose(np.linalg.norm(g["s2_pursuer_pos"], axis=-1), 1.0, atol=1e-12)
    assert (np.abs(g["s1_pos"]) <= 1).all() and g["s1_pos"].std() > 0.5


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_level4_position_is_the_reference_mapping_at_every_radius(g, prec):
    """generate_positions on free u at r below, at and above min_z (the phi range switches at r >= min_z, exp03_vFinal_task.py:591-601)."""
    from oracle import te_oracle as O
    for k, r in enumerate(g["l4_free_r"]):
        for j in range(g["l4_free_u"].shape[1]):
            got = O.level4_position(float(r), 4.0, g["l4_free_u"][k, j, 0], g["l4_free_u"][k, j, 1], prec)
            np.testing.assert_allclose(got, g["l4_free_pos"][k, j], rtol=0, atol=1e-12 if prec == "f64" else 2e-6 * max(1.0, r))


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_oracle_spawns_where_the_reference_sampler_would(g, prec):
    eng = _engine(prec)
    assert SP.replay_exp03(g, eng) == int(g["n_envs"]) * (3 + sum(range(2, 10)))
    assert SP.replay_stage02(g, eng) == int(g["n_envs"]) * (10 + 8 * len(g["s2_respawn_steps"]))
    assert SP.replay_stage01(g, eng) == int(g["n_envs"]) * (3 + len(g["s1_catch_steps"]))
