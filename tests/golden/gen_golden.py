#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the reference's own importable
modules.  Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

/root/reference never travels to the GPU box; only the small .npz/.json outputs do.  Nothing here
copies reference source: the modules are imported (or loaded by file path where their package
__init__ would pull pybullet/gymnasium) and called on seeded random inputs.

Outputs (inputs + expected outputs, all plain numeric arrays):
  lidar_math.npz     LidarMath conversions, binning, add_features (closer/farther wins), LIDARSpec shapes
  gun.npz            Gun state-machine traces
  kamikaze.npz       KamikazeNavigator (both variants) state + command traces
  geometry.npz       GeometryUtils cone / angle
  normalization.npz  normalize_inertial_data
  snapshot_buffer.npz  lidar_buffer.SnapshotBuffer / LiDARBufferManager on a scripted publication history
  transform_features.npz  LidarMath.transform_features / neighbor_sphere_from_new_frame on 256 snapshot pairs (pybullet.rotateVector = a numpy stand-in)
  ref_level5_obs.npz decoded io_data0.h5 recorded observations (the only PyBullet-made numbers in the tree)
"""
import importlib.util
import os
import struct
import sys

import numpy as np

REF = "/root/reference/src"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)


def by_path(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def gen_lidar_math():
    from core.dataclasses.angle_grid import LIDARSpec
    from core.entities.entity_type import EntityType

    lm = by_path("ref_lidar_math", "core/entities/quadcopters/components/sensors/components/lidar_math.py")
    spec = LIDARSpec(theta_initial_radian=0, theta_final_radian=np.pi, phi_initial_radian=-np.pi,
                     phi_final_radian=np.pi, resolution=16, n_channels=3, max_radius=40.0)
    math = lm.LidarMath(spec)
    rng = np.random.RandomState(20251004)
    vecs = rng.uniform(-50, 50, (1000, 3))
    # poles, seam, zero, beyond max radius, axis-aligned
    special = np.array([[0, 0, 1], [0, 0, -1], [0, 0, 0], [-1, 0, 0], [-1, 1e-12, 0], [-1, -1e-12, 0], [1, 0, 0],
                        [0, 1, 0], [0, -1, 0], [100, 0, 0], [0, 0, 39.9999], [1e-9, 0, 0], [30, 30, 30]], float)
    vecs[: len(special)] = special
    sph = np.array([lm.LidarMath.cartesian_to_spherical(v) for v in vecs])
    back = np.array([lm.LidarMath.spherical_to_cartesian(s) for s in sph])
    th_idx = np.array([math.theta_index_from_radian(s[1]) for s in sph], np.int32)
    ph_idx = np.array([math.phi_index_from_radian(s[2]) for s in sph], np.int32)
    nd = np.array([math.normalize_distance(s[0]) for s in sph])
    # direct angle sweeps incl. the exact ends
    thetas = np.concatenate([np.linspace(0, np.pi, 257), [np.pi / 2, np.pi, 0.0]])
    phis = np.concatenate([np.linspace(-np.pi, np.pi, 513), [0.0, np.pi, -np.pi]])
    th_sweep = np.array([math.theta_index_from_radian(t) for t in thetas], np.int32)
    ph_sweep = np.array([math.phi_index_from_radian(p) for p in phis], np.int32)
    th_center = np.array([math.theta_radian_from_index(i) for i in range(13)])
    ph_center = np.array([math.phi_radian_from_index(i) for i in range(26)])

    # add_features: 50 lists with deliberate collisions; entity types mixed (enum -> value/5, float kept)
    feats_all, n_feats, closer, farther = [], [], [], []
    types = [EntityType.LOITERINGMUNITION, EntityType.LOYALWINGMAN]
    for k in range(50):
        n = rng.randint(1, 12)
        f = np.zeros((12, 5))
        base_theta = rng.uniform(0, np.pi, n)
        base_phi = rng.uniform(-np.pi, np.pi, n)
        for i in range(1, n):  # force ~40 % of features into the previous feature's cell
            if rng.rand() < 0.4:
                base_theta[i] = base_theta[i - 1] + rng.uniform(-1e-3, 1e-3)
                base_phi[i] = base_phi[i - 1] + rng.uniform(-1e-3, 1e-3)
        base_theta = np.clip(base_theta, 0, np.pi)
        base_phi = np.clip(base_phi, -np.pi, np.pi)
        r = rng.uniform(0, 1.2, n).clip(0, 1)
        tsel = rng.randint(0, 2, n)
        feature_list_enum, feature_list_float = [], []
        for i in range(n):
            et = types[tsel[i]]
            feature_list_enum.append((r[i], base_theta[i], base_phi[i], et, 0.1, 7 + i))
            f[i] = [r[i], base_theta[i], base_phi[i], et.value / 5, 0.1]
        s1, _ = math.add_features(spec.empty_sphere(), feature_list_enum)
        s2, _ = math.add_features(spec.empty_sphere(), feature_list_enum, invert_prioritization_criteria=True)
        feats_all.append(f); n_feats.append(n); closer.append(s1.copy()); farther.append(s2.copy())
    shapes = []
    for res in (1, 4, 8, 16):
        sp = LIDARSpec(theta_initial_radian=0, theta_final_radian=np.pi, phi_initial_radian=-np.pi,
                       phi_final_radian=np.pi, resolution=res, n_channels=3, max_radius=1.0)
        shapes.append([res, *sp.shape, *sp.stacked_sphere_shape(5)])
    # extract_features round trip on the first closer-wins sphere
    ext = np.array(math.extract_features(closer[0]))
    np.savez_compressed(os.path.join(OUT, "lidar_math.npz"), vecs=vecs, sph=sph, back=back, th_idx=th_idx,
                        ph_idx=ph_idx, norm_dist=nd, max_radius=40.0, thetas=thetas, phis=phis, th_sweep=th_sweep,
                        ph_sweep=ph_sweep, th_center=th_center, ph_center=ph_center,
                        feats=np.array(feats_all), n_feats=np.array(n_feats, np.int32),
                        closer=np.array(closer, np.float32), farther=np.array(farther, np.float32),
                        shapes=np.array(shapes, np.int32), extract0=ext)


def gen_gun():
    gun_mod = by_path("ref_gun", "core/entities/quadcopters/components/weapons/gun.py")
    rng = np.random.RandomState(7)

    class Draw:  # stands in for the `random` module inside gun.py: deterministic per-event draw
        value = 0.0

        @classmethod
        def random(cls):
            return cls.value

    gun_mod.random = Draw
    traces = []
    for munition in (0, 1, 4, 20):
        g = gun_mod.Gun(parent_id=1)
        g.set_munition(munition)
        g.reset()
        n = 200
        steps = np.cumsum(rng.randint(0, 4, n)) + 1  # broadcast steps (non-decreasing, gaps)
        shoot = (rng.rand(n) < 0.5).astype(np.int32)
        draws = rng.rand(n)
        draws[rng.rand(n) < 0.1] = 0.9  # exactly at the hit threshold -> miss (>=)
        hit = np.zeros(n, np.int32); mun = np.zeros(n, np.int32); st = np.zeros((n, 3))
        for i in range(n):
            g._subscriber_simulation_step({"step": int(steps[i]), "timestep": 1 / 15}, 0)
            Draw.value = float(draws[i])
            hit[i] = int(g.shoot()) if shoot[i] else 0
            mun[i] = g.munition
            st[i] = g.get_state()
        traces.append(dict(munition=munition, steps=steps.astype(np.int32), shoot=shoot, draws=draws, hit=hit, mun=mun,
                           state=st))
    g = gun_mod.Gun(parent_id=1)
    np.savez_compressed(os.path.join(OUT, "gun.npz"), cooldown=float(g.cooldown_steps), hit_prob=g.fire_probability,
                        initial_state=g.get_state(),
                        **{f"{k}_{t['munition']}": v for t in traces for k, v in t.items() if k != "munition"})


def gen_kamikaze():
    from core.entities.navigators import loitering_munition_navigator as general
    from core.entities.navigators import loitering_munition_navigator_air_combat_only as aco

    rng = np.random.RandomState(11)
    STATE = {"WaitState": 0, "CollideWithWingman": 1, "CollideWithBuilding": 2}

    class Agent:
        def __init__(self, id_):
            self.id = id_
            self.inertial_data = {"position": np.zeros(3)}
            self.cmd = None

        def drive(self, c):
            self.cmd = np.array(c, float)

    class Offsets:
        """Stub exposing the three queries the navigators use (offsets_handler.py:228-254,427-434)."""

        def __init__(self):
            self.p_ids, self.p_pos, self.inv = [], np.zeros((0, 3)), {}

        def get_purusers_positions(self):
            return self.p_pos

        def identify_closest_pursuer(self, invader_id):
            if invader_id not in self.inv or len(self.p_ids) == 0:
                return -1
            d = np.linalg.norm(self.p_pos - self.inv[invader_id], axis=1)
            return self.p_ids[int(np.argmin(d))]

        def get_pursuer_position(self, pid):
            return self.p_pos[self.p_ids.index(pid)]

    out = {}
    P, I, T = 2, 3, 40
    for name, mod, cone in (("aco", aco, 0), ("general", general, 1)):
        building = np.array([0.0, 0.0, 0.1]) if cone == 0 else np.array([0.0, 0.0, 1.0])
        nav = mod.KamikazeNavigator(building)
        agents = [Agent(100 + j) for j in range(I)]
        pos = np.zeros((T, P + I, 3)); mask = np.zeros(T, np.uint32)
        st_in = np.zeros((T, P + I), np.int32); st_out = np.zeros((T, P + I), np.int32); cmd = np.zeros((T, P + I, 4))
        off = Offsets()
        cur_state = {a.id: 0 for a in agents}
        for t in range(T):
            pos[t, :P] = rng.uniform(-6, 6, (P, 3)) * [1, 1, 0.5] + [0, 0, 3]
            pos[t, P:] = rng.uniform(-8, 8, (I, 3)) * [1, 1, 0.3] + [0, 0, 6]
            if t % 7 == 3:  # coincident invader/pursuer -> zero vector branch
                pos[t, P] = pos[t, 0]
            m = 0
            # all pursuers dead for a while (air-combat-only variant only: the general navigator raises there,
            # loitering_munition_navigator.py:92-94 -> offsets_handler.py:430 index(-1))
            alive_p = [p for p in range(P) if not (cone == 0 and t in range(20, 26)) and not (t % 5 == 4 and p == 1)]
            for p in alive_p:
                m |= 1 << p
            alive_i = [j for j in range(I) if not (t % 6 == 5 and j == 2)]
            for j in alive_i:
                m |= 1 << (P + j)
            mask[t] = m
            if t == 30:  # navigator.reset(): everybody back to Wait
                nav.reset(); cur_state = {a.id: 0 for a in agents}
            off.p_ids = [1 + p for p in alive_p]
            off.p_pos = pos[t, alive_p].reshape(-1, 3)
            off.inv = {agents[j].id: pos[t, P + j] for j in alive_i}
            for j in alive_i:
                a = agents[j]
                a.inertial_data = {"position": pos[t, P + j].copy()}
                st_in[t, P + j] = cur_state[a.id]
                nav.update(a, off)
                cur_state[a.id] = STATE[nav.fetch_state(a).name]
                st_out[t, P + j] = cur_state[a.id]
                cmd[t, P + j] = a.cmd
        out.update({f"{name}_pos": pos, f"{name}_mask": mask, f"{name}_state_in": st_in, f"{name}_state_out": st_out,
                    f"{name}_cmd": cmd, f"{name}_building": building, f"{name}_speed": nav.velocity})
    np.savez_compressed(os.path.join(OUT, "kamikaze.npz"), P=P, I=I, **out)


def gen_geometry():
    from core.entities.navigators.geometry_utils import GeometryUtils

    rng = np.random.RandomState(5)
    pts = rng.uniform(-5, 5, (500, 3)); apex = rng.uniform(-5, 5, (500, 3)); base = rng.uniform(-5, 5, (500, 3))
    deg = rng.choice([45.0, 60.0, 90.0], 500)
    inside = np.array([GeometryUtils.is_point_inside_cone(p, a, b, d) for p, a, b, d in zip(pts, apex, base, deg)],
                      np.int32)
    ang = np.array([GeometryUtils.degrees_between_vectors(p - a, b - a) for p, a, b in zip(pts, apex, base)])
    np.savez_compressed(os.path.join(OUT, "geometry.npz"), pts=pts, apex=apex, base=base, deg=deg, inside=inside,
                        angle=ang)


def gen_normalization():
    norm = by_path("ref_norm", "threatengage/environments/level4/components/utils/normalization.py")
    rng = np.random.RandomState(3)
    n = 200
    pos = rng.uniform(-30, 30, (n, 3)); vel = rng.uniform(-5, 5, (n, 3))
    att = rng.uniform(-np.pi, np.pi, (n, 3)); rate = rng.uniform(-10, 10, (n, 3))
    out = np.zeros((n, 12), np.float32)
    for i in range(n):
        d = norm.normalize_inertial_data(dict(position=pos[i].copy(), velocity=vel[i].copy(), attitude=att[i].copy(),
                                              angular_rate=rate[i].copy()), 10 * 1000 / 3600, 20.0)
        out[i] = np.concatenate([d["position"], d["velocity"], d["attitude"], d["angular_rate"]])
    np.savez_compressed(os.path.join(OUT, "normalization.npz"), pos=pos, vel=vel, att=att, rate=rate, out=out,
                        max_speed=10 * 1000 / 3600, dome_radius=20.0)


def gen_snapshot_buffer():
    """SnapshotBuffer / LiDARBufferManager (lidar_buffer.py:10-157,262-517) driven with a scripted publication history:
    which snapshot `get_snapshot(publisher, delta)` returns at every step, its normalized_delta, who is a candidate
    of `get_random_neighborhood`, and the (inclusive) range its ages are drawn from."""
    from core.dataclasses.message_context import MessageContext
    from core.entities.entity_type import EntityType
    from core.notification_system.topics_enum import TopicsEnum

    lb = by_path("ref_lidar_buffer", "core/entities/quadcopters/components/sensors/components/lidar_buffer.py")
    T, P, I = 26, 6, 2
    death = {1: 7, 3: 15, 4: 15, 5: 21}           # wingman -> step at which it is disarmed (publishes up to death - 1)
    born = {p: 1 for p in range(P)}
    mgr = lb.LiDARBufferManager(current_step=0, max_buffer_size=10)
    ids = {p: 100 + p for p in range(P)}
    inv_ids = {j: 200 + j for j in range(I)}
    found = np.zeros((T + 1, P, 10), np.int32)     # payload (publication step) or 0
    ndelta = np.zeros((T + 1, P, 10), np.float64)
    cands = np.zeros((T + 1, P), np.uint8)
    age_range = np.zeros((T + 1, 2), np.int32)
    captured = {}

    class Rand:  # stands in for the `random` module inside lidar_buffer.py: records instead of drawing
        @staticmethod
        def sample(population, k):
            captured["candidates"] = list(population)
            return list(population)[:k]

        @staticmethod
        def randint(a, b):
            captured["range"] = (a, b)
            return a

    lb.random = Rand
    for t in range(1, T + 1):
        mgr.update_current_step(t)  # AGENT_STEP_BROADCAST (base_lidar.py:62-66)
        for p in range(P):
            if t == death.get(p, 10 ** 9):
                mgr.close_buffer(ids[p], TopicsEnum.INERTIAL_DATA_BROADCAST)  # messageHub.terminate on disarm
            if born[p] <= t < death.get(p, 10 ** 9):
                ctx = MessageContext(publisher_id=ids[p], step=t - 1, entity_type=EntityType.LOYALWINGMAN)
                mgr.buffer_message({"position": [float(t), 0.0, 0.0]}, ctx, TopicsEnum.INERTIAL_DATA_BROADCAST)
                mgr.buffer_message({"features": [t]}, ctx, TopicsEnum.LIDAR_DATA_BROADCAST)
        for j in range(I):
            ctx = MessageContext(publisher_id=inv_ids[j], step=t - 1, entity_type=EntityType.LOITERINGMUNITION)
            mgr.buffer_message({"position": [0.0, float(t), 0.0]}, ctx, TopicsEnum.INERTIAL_DATA_BROADCAST)
        for p in range(P):
            for d in range(1, 10):
                snap = mgr._buffer.get_snapshot(ids[p], d)
                if snap is not None and snap.lidar_features is not None:
                    found[t, p, d] = snap.lidar_features[0]
                    ndelta[t, p, d] = snap.normalized_delta
        captured.clear()
        mgr.get_random_neighborhood(n_neighbors=99)
        for pid in captured.get("candidates", []):
            cands[t, pid - 100] = 1
        age_range[t] = captured.get("range", (0, 0))
    np.savez_compressed(os.path.join(OUT, "snapshot_buffer.npz"), found=found, normalized_delta=ndelta, candidates=cands,
                        age_range=age_range, death=np.array([[k, v] for k, v in death.items()], np.int32), T=T)
    print("snapshot_buffer: lookups", int((found > 0).sum()), "of", found.size)


def _rotate_vector_xyzw(q, v):
    """The ONE primitive of transform_features that lives in pybullet: rotateVector(quaternion xyzw, vector) = q (0, v) q^-1 (sandwich
    product, unit quaternion).  Stand-in for `pybullet.rotateVector` only; checked against analytic cases in gen_transform_features."""
    x, y, z, w = (float(c) for c in q)
    u, s = np.array([x, y, z]), w
    v = np.asarray(v, float)
    return tuple(2.0 * np.dot(u, v) * u + (s * s - np.dot(u, u)) * v + 2.0 * s * np.cross(u, v))


def gen_transform_features():
    """LidarMath.transform_features / reframe / neighbor_sphere_from_new_frame (lidar_math.py:53-83,186-260,324-352) — the COMPOSITION that
    re-projects a neighbour's snapshot into the observer's frame — run by the reference on 256 (neighbour, own) snapshot pairs.  The only
    thing the reference does not do itself here is the rotation primitive: `pybullet.rotateVector` is a stand-in (6 lines of numpy above,
    asserted on identity, quarter turns about each axis, a composition and the xyzw order).  "Composition pinned modulo the rotation primitive"."""
    import types

    from core.dataclasses.angle_grid import LIDARSpec
    from core.dataclasses.perception_snapshot import PerceptionSnapshot
    from core.entities.entity_type import EntityType
    from core.notification_system.topics_enum import TopicsEnum

    # --- the stand-in against closed forms
    h = np.sqrt(0.5)
    R = _rotate_vector_xyzw
    assert np.allclose(R([0, 0, 0, 1], [1, 2, 3]), [1, 2, 3])
    assert np.allclose(R([0, 0, h, h], [1, 0, 0]), [0, 1, 0])          # +90 deg about z: x -> y
    assert np.allclose(R([h, 0, 0, h], [0, 1, 0]), [0, 0, 1])          # +90 deg about x: y -> z
    assert np.allclose(R([0, h, 0, h], [0, 0, 1]), [1, 0, 0])          # +90 deg about y: z -> x
    assert np.allclose(R([0, 0, h, h], R([h, 0, 0, h], [0, 1, 0])), [0, 0, 1])   # composition: (x turn) then (z turn) leaves z alone
    assert np.allclose(R([0, 0, 1, 0], [1, 0, 0]), [-1, 0, 0])         # xyzw order: (0,0,1,0) is a half turn about z, not the identity
    pb = types.ModuleType("pybullet")
    pb.rotateVector = _rotate_vector_xyzw

    def tripwire(*a, **k):
        raise AssertionError("only rotateVector may be reached")
    pb.getMatrixFromQuaternion = tripwire
    sys.modules["pybullet"] = pb
    try:
        lm = by_path("ref_lidar_math_tf", "core/entities/quadcopters/components/sensors/components/lidar_math.py")
        # the reference's own KAT (math_test.py:11-69; its PerceptionSnapshot call passes a max_delta_step the dataclass no longer takes)
        spec20 = LIDARSpec(theta_initial_radian=0, theta_final_radian=np.pi, phi_initial_radian=-np.pi, phi_final_radian=np.pi,
                           resolution=16, n_channels=3, max_radius=20)
        m20 = lm.LidarMath(spec20)
        imu = TopicsEnum.INERTIAL_DATA_BROADCAST; lid = TopicsEnum.LIDAR_DATA_BROADCAST
        nb = PerceptionSnapshot(topics={imu: {"position": [0.0, 0.0, 0.0], "quaternion": [0.0, 0.0, 0.0, 1.0]},
                                        lid: {"features": [(0.5, np.pi / 2, 0.0, EntityType.LOITERINGMUNITION, 0, 1)]}},
                                publisher_id=1, step=0, entity_type=EntityType.LOYALWINGMAN)
        own = PerceptionSnapshot(topics={imu: {"position": [1.0, 0.0, 0.0], "quaternion": [0.0, 0.0, 0.0, 1.0]}}, publisher_id=2, step=0,
                                 entity_type=EntityType.LOYALWINGMAN)
        (kr, kt, kp, _, _), = m20.transform_features(nb, own)
        assert np.isclose(kr, 0.45, atol=1e-2) and np.isclose(kt, np.pi / 2, atol=1e-2) and np.isclose(kp, 0.0, atol=1e-2)

        spec = LIDARSpec(theta_initial_radian=0, theta_final_radian=np.pi, phi_initial_radian=-np.pi, phi_final_radian=np.pi,
                         resolution=16, n_channels=3, max_radius=40.0)
        math = lm.LidarMath(spec)
        rng = np.random.RandomState(20251005)
        N, F, P, D = 256, 11, 2, 12                      # level5_c1's shape: 2 wingmen, 12 drones, at most D - 1 features per snapshot
        DELTA = 0.3                                      # the neighbour snapshot's normalized_delta (set on retrieval, lidar_buffer.py:143-150)
        nb_pos = np.zeros((N, 3), np.float32); nb_quat = np.zeros((N, 4), np.float32)
        own_pos = np.zeros((N, 3), np.float32); own_quat = np.zeros((N, 4), np.float32)
        n_feat = np.zeros(N, np.int32); feats = np.zeros((N, F, 4), np.float64)       # r_hat, theta, phi, publisher slot
        n_out = np.zeros(N, np.int32); out = np.zeros((N, F, 5), np.float64)
        spheres = np.ones((N, 3, 13, 26), np.float32)

        def quat(i):
            if i % 2:   # a flying attitude: small roll / pitch, any yaw
                r, p_, y = rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), rng.uniform(-np.pi, np.pi)
                cr, sr, cp, sp, cy, sy = np.cos(r / 2), np.sin(r / 2), np.cos(p_ / 2), np.sin(p_ / 2), np.cos(y / 2), np.sin(y / 2)
                return np.array([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy])
            q = rng.normal(size=4)
            return q / np.linalg.norm(q)
        self_echoes = collisions = clipped = 0
        for i in range(N):
            nb_pos[i] = rng.uniform(-6, 6, 3); own_pos[i] = rng.uniform(-6, 6, 3)
            nb_quat[i] = quat(i); own_quat[i] = quat(i + 1)
            if i == 0:  # the KAT's geometry at this spec's radius: target 20 m ahead of a neighbour at the origin, observer at x = 1
                nb_pos[i] = 0; own_pos[i] = (1, 0, 0); nb_quat[i] = own_quat[i] = (0, 0, 0, 1)
            k = int(rng.randint(0, F + 1)) if i else 1
            slots = rng.permutation(np.r_[0, 2:D])[:k]     # publishers: the observer itself (slot 0: self echo) or an invader; never the neighbour
            fl = []
            for j, sl in enumerate(slots):
                r, th, ph = rng.uniform(0.02, 0.45), rng.uniform(0, np.pi), rng.uniform(-np.pi, np.pi)
                if i == 0: r, th, ph, sl = 0.5, np.pi / 2, 0.0, 5
                if j and rng.rand() < 0.3:                 # a second target almost behind an earlier one: same cell from afar -> farther wins
                    r0, th, ph = fl[0][0], fl[0][1] + rng.uniform(-0.01, 0.01), fl[0][2] + rng.uniform(-0.01, 0.01)
                    r = r0 * rng.uniform(0.5, 1.6); th = float(np.clip(th, 0, np.pi))
                if rng.rand() < 0.08: r = rng.uniform(0.8, 1.0)   # beyond the observer's range after the shift: r_hat clips to 1
                et = EntityType.LOYALWINGMAN if sl < P else EntityType.LOITERINGMUNITION
                fl.append((float(r), float(th), float(ph), et, 0.0, int(sl)))
                feats[i, j] = (r, th, ph, sl)
            n_feat[i] = k
            nb = PerceptionSnapshot(topics={imu: {"position": nb_pos[i].tolist(), "quaternion": nb_quat[i].tolist()}, lid: {"features": fl}},
                                    publisher_id=1, step=5, entity_type=EntityType.LOYALWINGMAN, normalized_delta=DELTA)
            own = PerceptionSnapshot(topics={imu: {"position": own_pos[i].tolist(), "quaternion": own_quat[i].tolist()}}, publisher_id=0, step=7,
                                     entity_type=EntityType.LOYALWINGMAN)
            tf = math.transform_features(nb, own)
            n_out[i] = len(tf)
            for j, t in enumerate(tf):
                out[i, j] = (t[0], t[1], t[2], t[3].value / 5, t[4])
            self_echoes += k - len(tf); clipped += sum(t[0] >= 1.0 for t in tf)
            sph = math.neighbor_sphere_from_new_frame(nb, own)
            assert sph.shape == (3, 13, 26)
            spheres[i] = sph
            collisions += len(tf) - int((sph[0] < 1).sum()) - sum(t[0] >= 1.0 for t in tf)
            assert np.all(sph[2][sph[0] < 1] == DELTA)
        assert abs(out[0, 0, 0] - 19 / 40) < 1e-6
    finally:
        del sys.modules["pybullet"]
    np.savez_compressed(os.path.join(OUT, "transform_features.npz"), nb_pos=nb_pos, nb_quat=nb_quat, own_pos=own_pos, own_quat=own_quat,
                        n_feat=n_feat, feats=feats, n_out=n_out, out=out, spheres=spheres, delta=DELTA)
    print("transform_features: pairs", N, "features", int(n_feat.sum()), "self echoes skipped", self_echoes, "cells lost to farther-wins", collisions,
          "r_hat clipped to 1", clipped, "hit cells", int((spheres[:, 0] < 1).sum()))


def gen_h5_fixture():
    """Decode src/core/rl_framework/utils/output/collect_and_save/io_data0.h5 without h5py
    (HDF5 v1 B-tree chunk index, one uncompressed chunk per sample)."""
    path = os.path.join(REF, "core/rl_framework/utils/output/collect_and_save/io_data0.h5")
    raw = open(path, "rb").read()

    def chunks(node_off, ndims):
        assert raw[node_off:node_off + 4] == b"TREE", node_off
        ntype, level, entries = struct.unpack_from("<BBH", raw, node_off + 4)
        assert ntype == 1 and level == 0
        off = node_off + 8 + 16
        res = []
        for _ in range(entries):
            size, _fm = struct.unpack_from("<II", raw, off); off += 8
            offs = struct.unpack_from("<" + "Q" * (ndims + 1), raw, off); off += 8 * (ndims + 1)
            (addr,) = struct.unpack_from("<Q", raw, off); off += 8
            res.append((offs, size, addr))
        return res

    def dataset(node_off, sample_shape, chunk_shape):
        nd = 1 + len(sample_shape)
        ch = chunks(node_off, nd)
        n = max(o[0] for o, _, _ in ch) + 1
        arr = np.zeros((n, *sample_shape), np.float32)
        for offs, size, addr in ch:
            blk = np.frombuffer(raw, "<f4", size // 4, addr).reshape(chunk_shape)
            idx = tuple(slice(o, o + c) for o, c in zip(offs[:nd], chunk_shape))
            arr[idx] = blk
        return arr

    inertial = dataset(49592, (15,), (1, 15))
    last_action = dataset(52480, (4,), (1, 4))
    teacher_actions = dataset(55368, (4,), (1, 4))
    spheres = dataset(17920, (6, 3, 13, 26), (1, 3, 3, 13, 26))
    ch = chunks(46704, 2)
    mask = np.zeros((len(ch), 6), np.uint8)
    for offs, size, addr in ch:
        mask[offs[0]] = np.frombuffer(raw, np.uint8, size, addr)
    # sparse hits: (sample, sphere, theta, phi, r_hat, flag, time)
    hit = np.argwhere(spheres[:, :, 0] < 1)
    vals = np.array([[*h, spheres[h[0], h[1], 0, h[2], h[3]], spheres[h[0], h[1], 1, h[2], h[3]],
                      spheres[h[0], h[1], 2, h[2], h[3]]] for h in hit], np.float64)
    np.savez_compressed(os.path.join(OUT, "ref_level5_obs.npz"), inertial=inertial, last_action=last_action,
                        teacher_actions=teacher_actions, mask=mask, hits=vals,
                        empty_fraction=float((spheres == 1).mean()))
    print("h5: samples", inertial.shape[0], "hits", len(vals), "mask rows", mask.sum(1))


if __name__ == "__main__":
    gen_lidar_math(); print("lidar_math ok")
    gen_gun(); print("gun ok")
    gen_kamikaze(); print("kamikaze ok")
    gen_geometry(); print("geometry ok")
    gen_normalization(); print("normalization ok")
    gen_snapshot_buffer()
    gen_transform_features()
    gen_h5_fixture()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
