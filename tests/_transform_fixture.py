"""Replay of tests/golden/transform_features.npz (the reference's LidarMath.transform_features / neighbor_sphere_from_new_frame run on 256
snapshot pairs, tests/golden/gen_golden.py:gen_transform_features) through the stacked observation of a level5_c1 environment
(2 wingmen, 12 drones): env i holds pair i — the observer's snapshots in the ring of wingman 0, the neighbour's (pose + features) in the
ring of wingman 1, the same content under every stamp of the last ten steps, so that whichever neighbour and age the observation draws,
the snapshot it re-projects is the fixture's.  Shared by the oracle test (CPU) and the C-ABI test on the GPU."""
import os

import numpy as np

from dronechase_amd import config as K
from tests._blob import Blob

FX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "transform_features.npz")
STEP = 12
TOL = 5e-6            # r_hat of a re-projected feature: float32 arithmetic on poses up to 12 m apart, ranges up to 40 m (the reference: float64)
CELL_MARGIN = 5e-5    # rad: closer than this to a cell edge a float32 angle may legitimately fall into the neighbouring cell


def load():
    return np.load(FX)


def build_state(cfg, words, fx):
    """words = the state blob of a freshly reset level5_c1 env of 256 envs -> the blob with the fixture's snapshots in the rings."""
    N, D, P = int(cfg.n_envs), int(cfg.n_drones), int(cfg.n_pursuers)
    assert (N, D, P) == (256, 12, 2)
    b = Blob(words, N, D)
    ew = K.ring_entry_words(D)
    ring = b.w[N * (D * K.DRONE_WORDS + K.ENV_WORDS):].reshape(N, P, K.RING_DEPTH, ew)
    ring[:] = 0
    f32 = lambda a: np.asarray(a, np.float32).view(np.uint32)
    for e in range(N):
        b.place(e, 0, fx["own_pos"][e]); b.set_f(e, 0, "QUAT", fx["own_quat"][e])
        b.place(e, 1, fx["nb_pos"][e]); b.set_f(e, 1, "QUAT", fx["nb_quat"][e])
        for d in range(2, D):
            b.set_i(e, d, "ARMED", 0)
        b.set_ei(e, "STEP", STEP)
        b.refresh_snapshot(e)
        for s in range(STEP - 9, STEP + 1):
            own = ring[e, 0, s % K.RING_DEPTH]; nb = ring[e, 1, s % K.RING_DEPTH]
            own[0] = s; own[1] = 0; own[2:5] = f32(fx["own_pos"][e]); own[5:9] = f32(fx["own_quat"][e])
            n = int(fx["n_feat"][e])
            nb[0] = s; nb[1] = n; nb[2:5] = f32(fx["nb_pos"][e]); nb[5:9] = f32(fx["nb_quat"][e])
            for j in range(n):
                r, th, ph, slot = fx["feats"][e, j]
                typ = 3 if slot < P else 1                                   # TE_TYPE_LOYALWINGMAN / TE_TYPE_LOITERINGMUNITION (entity_type.py)
                nb[12 + 4 * j: 15 + 4 * j] = f32([r, th, ph]); nb[15 + 4 * j] = np.uint32(typ | (int(slot) << 8))
    return b


def ambiguous(fx):
    """Envs where a float32 rounding may legitimately change the sphere: a re-projected feature within CELL_MARGIN of a cell edge, two
    features of one cell whose ranges tie, or a range at the clip."""
    N = len(fx["n_out"])
    amb = np.zeros(N, bool)
    for e in range(N):
        o = fx["out"][e, : fx["n_out"][e]]
        if not len(o):
            continue
        ft = o[:, 1] / (np.pi / K.LIDAR_NTHETA); fp = (o[:, 2] + np.pi) / (2 * np.pi / K.LIDAR_NPHI)
        dth = np.abs(ft - np.round(ft)) * (np.pi / K.LIDAR_NTHETA); dph = np.abs(fp - np.round(fp)) * (2 * np.pi / K.LIDAR_NPHI)
        inner_t = (np.round(ft) > 0) & (np.round(ft) < K.LIDAR_NTHETA)      # the outer edges clip, they do not flip
        inner_p = (np.round(fp) > 0) & (np.round(fp) < K.LIDAR_NPHI)
        if (dth[inner_t] < CELL_MARGIN).any() or (dph[inner_p] < CELL_MARGIN).any():
            amb[e] = True
        cells = np.minimum(ft.astype(int), K.LIDAR_NTHETA - 1) * K.LIDAR_NPHI + np.minimum(fp.astype(int), K.LIDAR_NPHI - 1)
        for c in np.unique(cells):
            r = np.sort(o[cells == c, 0])
            if len(r) > 1 and np.min(np.diff(r)) < 1e-5:
                amb[e] = True
        if (np.abs(o[:, 0] - 1.0) < 1e-5).any() and (o[:, 0] < 1.0).any() and ((o[:, 0] > 1 - 1e-5) & (o[:, 0] < 1)).any():
            amb[e] = True
    return amb


def check(cfg, fx, stacked, mask, episodes, draws_of):
    """stacked [256,6,3,13,26], mask [256,6] as te_observe_stacked / the oracle returned them; draws_of(env, episode) -> the oracle-independent
    record of the env's draws {n, who, age, perm} (Philox: both sides share it; pinned by the uniformity tests of tests/test_oracle_level5.py)."""
    N = 256
    amb = ambiguous(fx)
    seen_nb = seen_self = compared_cells = 0
    ones = np.ones((3, K.LIDAR_NTHETA, K.LIDAR_NPHI), np.float32)
    for e in range(N):
        d = draws_of(e, int(episodes[e]))
        nv = 1 + d["n"]
        stack = [ones]
        for who, age in zip(d["who"], d["age"]):
            if who == 0:
                stack.append(ones); seen_self += 1
            else:
                s = fx["spheres"][e].copy()
                s[2][s[2] == np.float32(fx["delta"])] = np.float32(age / 10.0)          # normalized_delta of the snapshot as retrieved (lidar_buffer.py:143-150)
                stack.append(s); seen_nb += 1
        for i in range(K.STACK_SPHERES):
            si = d["perm"][i]
            assert mask[e, i] == (1 if si < nv else 0), (e, i)
            want = stack[si] if si < nv else ones
            if amb[e]:
                continue
            got = stacked[e, i]
            assert np.abs(got - want).max() <= TOL, (e, i, float(np.abs(got - want).max()), d)
            compared_cells += int((want[0] < 1).sum())
    assert amb.sum() < 0.1 * N and seen_nb > 200 and seen_self > 100 and compared_cells > 500, (int(amb.sum()), seen_nb, seen_self, compared_cells)
    return dict(ambiguous=int(amb.sum()), neighbour_spheres=seen_nb, self_spheres=seen_self, hit_cells_compared=compared_cells)
