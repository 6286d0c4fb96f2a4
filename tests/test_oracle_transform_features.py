"""SURVEY.md 8 row a12, the piece round 2 left unpinned: the COMPOSITION LidarMath.transform_features -> add_features(invert) that
re-projects a neighbour's snapshot into the observer's frame (lidar_math.py:53-83,186-260,324-352).  tests/golden/transform_features.npz
holds the reference's own outputs on 256 snapshot pairs (its `pybullet.rotateVector` replaced by a 6-line numpy quaternion sandwich,
asserted on closed forms in the generator: "composition pinned modulo the rotation primitive").  Here: the oracle, both precisions;
tests/test_gpu_fixtures.py::test_transform_features_fixture_through_the_c_abi is the same replay through te_observe_stacked on the GPU."""
import numpy as np
import pytest

from oracle import te_oracle as O
from tests import _transform_fixture as T


def test_fixture_holds_the_references_own_kat():
    """math_test.py:11-69: a target 20 m ahead of a neighbour at the origin, seen by an observer 1 m along x: r_hat (20 - 1) / 40, theta pi / 2, phi 0."""
    fx = T.load()
    assert fx["n_out"][0] == 1
    np.testing.assert_allclose(fx["out"][0, 0, :3], [19 / 40, np.pi / 2, 0.0], atol=1e-9)
    assert fx["spheres"][0][0, 6, 13] == np.float32(19 / 40) and (fx["spheres"][0][0] < 1).sum() == 1
    assert (fx["n_out"] <= fx["n_feat"]).all() and (fx["n_out"] < fx["n_feat"]).sum() > 50      # self echoes of the observer are dropped


@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_oracle_reprojects_like_the_reference(precision):
    fx = T.load()
    cfg = O.default_config("level5_c1", n_envs=256, seed=11)
    o = O.OracleEnv(cfg, precision, threads=4)
    o.reset()
    b = T.build_state(cfg, o.get_state(), fx)
    o.set_state(b.w)
    stacked, mask, *_ = o.observe_stacked()
    episodes = [b.ei(e, "EPISODE") for e in range(256)]
    info = T.check(cfg, fx, stacked, mask, episodes, lambda e, ep: O.stack_draws(cfg, e, ep, T.STEP, 0b11))
    assert info["hit_cells_compared"] > 500
    o.close()
