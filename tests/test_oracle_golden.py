"""Pin the CPU oracle (oracle/te_oracle.c) against golden vectors produced by the reference's own
importable modules (tests/golden/gen_golden.py) and the known-answer scenarios of the reference's
tests.  CPU-only."""
import numpy as np
import pytest

from oracle import te_oracle as O

PREC = ["f64", "f32"]


def _tol(prec, t64, t32):
    return t64 if prec == "f64" else t32


@pytest.mark.parametrize("prec", PREC)
def test_cartesian_spherical_roundtrip(golden, prec):
    g = golden("lidar_math.npz")
    tol = _tol(prec, 1e-12, 2e-5)
    for v, s, b in zip(g["vecs"], g["sph"], g["back"]):
        got = O.vec_fn("ote_cartesian_to_spherical", v, 3, prec)
        # phi at the +-pi seam may flip sign under float rounding of y ~ 1e-12
        if abs(abs(s[2]) - np.pi) < 1e-6 and prec == "f32":
            assert abs(abs(got[2]) - np.pi) < 1e-5
            np.testing.assert_allclose(got[:2], s[:2], rtol=tol, atol=tol * 50)
        else:
            np.testing.assert_allclose(got, s, rtol=tol, atol=tol * 50)
        np.testing.assert_allclose(O.vec_fn("ote_spherical_to_cartesian", s, 3, prec), b, rtol=tol, atol=tol * 60)


def test_known_answers_spherical():
    # apps/threatengage_runner/stage02/auxiliary/test_lidar.py:12-20
    np.testing.assert_allclose(O.vec_fn("ote_spherical_to_cartesian", [1, np.pi / 4, np.pi / 4], 3),
                               [0.5, 0.5, np.sqrt(2) / 2], atol=1e-12)
    np.testing.assert_allclose(O.vec_fn("ote_cartesian_to_spherical", [0.5, 0.5, np.sqrt(2) / 2], 3),
                               [1, np.pi / 4, np.pi / 4], atol=1e-12)
    # SURVEY.md 8(c) smoke values
    assert O.theta_index(np.pi / 2) == 6 and O.phi_index(0.0) == 13 and O.phi_index(np.pi) == 25


def test_binning_exact_f64(golden):
    g = golden("lidar_math.npz")
    got_t = [O.theta_index(t) for t in g["thetas"]]
    got_p = [O.phi_index(p) for p in g["phis"]]
    assert got_t == list(g["th_sweep"]) and got_p == list(g["ph_sweep"])
    idx_t = [O.theta_index(s[1]) for s in g["sph"]]
    idx_p = [O.phi_index(s[2]) for s in g["sph"]]
    assert idx_t == list(g["th_idx"]) and idx_p == list(g["ph_idx"])
    nd = [O.normalize_distance(s[0], float(g["max_radius"])) for s in g["sph"]]
    np.testing.assert_allclose(nd, g["norm_dist"], atol=1e-15)


def test_binning_f32_off_boundary(golden):
    """In float32 a cell index may only differ where the angle sits within rounding of a cell edge."""
    g = golden("lidar_math.npz")
    bad = 0
    for s, ti, pi in zip(g["sph"], g["th_idx"], g["ph_idx"]):
        ft = s[1] / np.pi * 13
        fp = (s[2] + np.pi) / (2 * np.pi) * 26
        near = min(abs(ft - round(ft)), abs(fp - round(fp))) < 1e-4
        if O.theta_index(s[1], "f32") != ti or O.phi_index(s[2], "f32") != pi:
            assert near
            bad += 1
    assert bad <= 3


@pytest.mark.parametrize("prec", PREC)
def test_add_features_closer_and_farther(golden, prec):
    g = golden("lidar_math.npz")
    for f, n, c, far in zip(g["feats"], g["n_feats"], g["closer"], g["farther"]):
        np.testing.assert_array_equal(O.add_features(f[:n], False, prec), c)
        np.testing.assert_array_equal(O.add_features(f[:n], True, prec), far)


def test_lidar_spec_shapes(golden):
    g = golden("lidar_math.npz")
    row = {int(r[0]): r[1:] for r in g["shapes"]}
    assert list(row[16]) == [3, 13, 26, 6, 3, 13, 26]
    from dronechase_amd import config as K
    assert (K.LIDAR_CHANNELS, K.LIDAR_NTHETA, K.LIDAR_NPHI) == (3, 13, 26)
    # cell centres used by extract_features (lidar_math.py:103-105)
    np.testing.assert_allclose(g["th_center"], (np.arange(13) + 0.5) / 13 * np.pi)
    np.testing.assert_allclose(g["ph_center"], (np.arange(26) + 0.5) / 26 * 2 * np.pi - np.pi)


@pytest.mark.parametrize("prec", PREC)
@pytest.mark.parametrize("munition", [0, 1, 4, 20])
def test_gun_traces(golden, prec, munition):
    g = golden("gun.npz")
    assert float(g["cooldown"]) == 60.0 and float(g["hit_prob"]) == 0.9
    np.testing.assert_array_equal(g["initial_state"], [1, 0, 1])
    hit, mun, st = O.gun_trace(g[f"steps_{munition}"], g[f"shoot_{munition}"], g[f"draws_{munition}"], munition,
                               precision=prec)
    np.testing.assert_array_equal(hit, g[f"hit_{munition}"])
    np.testing.assert_array_equal(mun, g[f"mun_{munition}"])
    np.testing.assert_allclose(st, g[f"state_{munition}"], atol=1e-6 if prec == "f32" else 1e-15)


@pytest.mark.parametrize("prec", PREC)
@pytest.mark.parametrize("variant", ["aco", "general"])
def test_kamikaze_traces(golden, prec, variant):
    g = golden("kamikaze.npz")
    P, I = int(g["P"]), int(g["I"])
    pos, mask = g[f"{variant}_pos"], g[f"{variant}_mask"]
    st_in, st_out, cmd = g[f"{variant}_state_in"], g[f"{variant}_state_out"], g[f"{variant}_cmd"]
    tol = 1e-12 if prec == "f64" else 2e-6
    for t in range(len(pos)):
        nout, sp = O.kamikaze_scenario(P, I, pos[t], mask[t], st_in[t], float(g[f"{variant}_speed"]),
                                       cone_check=int(variant == "general"), building=g[f"{variant}_building"],
                                       precision=prec)
        for j in range(P, P + I):
            if not (int(mask[t]) >> j) & 1:
                continue
            assert nout[j] == st_out[t, j], (variant, t, j)
            c = cmd[t, j]
            n = np.linalg.norm(c[:3])
            d = c[:3] / (n if n > 0 else 1.0)
            want = np.array([c[3] * d[0], c[3] * d[1], 0.0, c[3] * d[2]])  # quadcopter.py:379-396
            np.testing.assert_allclose(sp[j], want, atol=tol)


def test_kamikaze_reference_kats():
    """core/entities/navigators/tests/test_lm_navigator.py:141-184 (general navigator, building (0,0,1))."""
    b = (0, 0, 1)
    pos = np.array([[0, 5, 5], [0, 1, 5], [0, 0, 10]], float)
    nout, _ = O.kamikaze_scenario(2, 1, pos, 0b111, [0, 0, 0], cone_check=1, building=b)
    assert nout[2] == 1  # CollideWithWingmanState
    pos = np.array([[0, 5, 5], [0, 10, 5], [0, 0, 10]], float)
    nout, _ = O.kamikaze_scenario(2, 1, pos, 0b111, [0, 0, 0], cone_check=1, building=b)
    assert nout[2] == 2  # CollideWithBuildingState


def test_wingman_reference_kats():
    """core/entities/navigators/tests/test_lw_navigator.py:140-212: gun available -> ChaseThreat;
    unavailable -> MoveToFormation; munition 0 -> SacrificeAttack (== chase)."""
    pos = np.array([[0, 0, 2], [1, 0, 2], [4, 0, 2], [0, 2, 2]], float)  # P=2, I=2
    form = np.array([-2.0, 0, 2])
    chase = O.wingman_scenario(2, 2, pos, form, 0b1111, 1, munition=5, last_fired=-60, step=3)
    np.testing.assert_allclose(chase, [-0.6 / np.sqrt(5), 0.6 * 2 / np.sqrt(5), 0, 0], atol=1e-12)  # closest invader (0,2,2)
    formation = O.wingman_scenario(2, 2, pos, form, 0b1111, 1, munition=5, last_fired=0, step=3)
    np.testing.assert_allclose(formation, [-0.6, 0, 0, 0], atol=1e-12)
    sacrifice = O.wingman_scenario(2, 2, pos, form, 0b1111, 1, munition=0, last_fired=0, step=3)
    np.testing.assert_allclose(sacrifice, chase, atol=1e-12)


@pytest.mark.parametrize("prec", PREC)
def test_geometry(golden, prec):
    g = golden("geometry.npz")
    mism = 0
    for p, a, b, d, ins, ang in zip(g["pts"], g["apex"], g["base"], g["deg"], g["inside"], g["angle"]):
        got = O.degrees_between(p - a, b - a, prec)
        np.testing.assert_allclose(got, ang, atol=1e-9 if prec == "f64" else 2e-2)
        if O.point_inside_cone(p, a, b, d, prec) != ins:
            assert prec == "f32" and abs(ang - d / 2) < 1e-2
            mism += 1
    assert mism <= 2
    # core/entities/navigators/legacy/geometry_utils_test.py:6-26
    assert abs(O.degrees_between([1, 0, 0], [0, 1, 0]) - 90) < 1e-9
    assert O.point_inside_cone([0, 0, -0.5], [0, 0, 0], [0, 0, -1], 60) == 1
    assert O.point_inside_cone([1, 1, 0], [0, 0, 0], [0, 0, -1], 60) == 0


@pytest.mark.parametrize("prec", PREC)
def test_normalization(golden, prec):
    g = golden("normalization.npz")
    for i in range(len(g["pos"])):
        got = O.normalize_inertial(g["pos"][i], g["vel"][i], g["att"][i], g["rate"][i], float(g["max_speed"]),
                                   float(g["dome_radius"]), prec)
        # te_config carries max_speed as float32 (2.7777777f): one float32 ulp vs the reference's double constant
        np.testing.assert_allclose(got, g["out"][i], rtol=0, atol=1.2e-7 if prec == "f64" else 2.4e-7)


def test_transform_features_reference_kat():
    """sensors/components/math_test.py:63-69: a feature at r_hat 0.5 (R=20), theta pi/2, phi 0 seen from a
    neighbour at the origin, own drone at (1,0,0), identity attitudes -> r_hat 0.45, theta pi/2, phi 0.
    The own-sphere path is the same reframe with local_vector = 0."""
    cart = O.vec_fn("ote_spherical_to_cartesian", [0.5 * 20, np.pi / 2, 0.0], 3)
    rel = O.rotate_vector([0, 0, 0, 1], cart - np.array([1.0, 0, 0]))
    sph = O.vec_fn("ote_cartesian_to_spherical", rel, 3)
    np.testing.assert_allclose([O.normalize_distance(sph[0], 20), sph[1], sph[2]], [0.45, np.pi / 2, 0], atol=1e-12)


def test_recorded_pybullet_observations(golden):
    """tests/golden/ref_level5_obs.npz = decoded io_data0.h5: the only numbers in
    the reference tree that PyBullet produced.  Pins the observation layout and a11 (own sphere):
    flags in {0.2, 0.6}, time plane 0.1, r_hat * 40 equals the distance between recorded wingman
    positions at the same step."""
    g = golden("ref_level5_obs.npz")
    inertial, hits, mask = g["inertial"], g["hits"], g["mask"]
    assert inertial.shape == (14, 15)
    np.testing.assert_allclose(inertial[:, 12:15], np.tile([1, 0, 1], (14, 1)))  # gun: full, no cooldown, available
    np.testing.assert_allclose(g["last_action"][:, 3], 0.6, atol=1e-7)            # BT speed (loyalwingman_navigator.py:37)
    np.testing.assert_array_equal(g["last_action"], g["teacher_actions"])
    assert set(np.round(hits[:, 5], 4)) <= {0.2, 0.6}
    np.testing.assert_allclose(hits[:, 6], 0.1, atol=1e-7)
    assert (mask.sum(1) >= 1).all()
    # Cross-sample consistency of the own sphere.  Which stack row is the OWN sphere is unknown
    # (the stack is shuffled, fused_lidar.py:246-262) but a sphere whose wingman hits reproduce every
    # other wingman's range must exist for every sample.
    from oracle.te_oracle import own_sphere_from_poses
    pos = inertial[:, 0:3].astype(np.float64) * 20.0
    eul = inertial[:, 6:9].astype(np.float64) * np.pi
    for half in (0, 7):
        P = 7
        for w in range(P):
            s = half + w
            sph = own_sphere_from_poses(pos[half:half + P], eul[s], w, np.ones(P, np.uint8), P, 40.0)
            mine = {(int(t), int(p)): (r, f) for t in range(13) for p in range(26)
                    for r, f in [(sph[0, t, p], sph[1, t, p])] if r < 1}
            best = 0
            for k in range(6):
                rec = {(int(h[2]), int(h[3])): (h[4], h[5]) for h in hits if int(h[0]) == s and int(h[1]) == k
                       and abs(h[5] - 0.6) < 1e-4 and abs(h[6] - 0.1) < 1e-6}
                same = sum(1 for c, (r, f) in mine.items() if c in rec and abs(rec[c][0] - r) < 2e-4)
                best = max(best, same)
            # invaders may occlude a wingman in the recording (closer wins), and the other wingmen are
            # recorded one pose later/earlier by at most one physics step: demand a clear majority.
            assert best >= max(1, len(mine) - 2), (s, best, len(mine))
