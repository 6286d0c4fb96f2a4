#!/usr/bin/env python3
"""Golden vectors for the TASK LOGIC of stage01 and stage02 (BASELINE configs 1 and 2), made by RUNNING the reference's own code:

  stage02   level3/components/stages.py            L3Stage1.on_step_middle / on_step_end (engagement on the stale matrix, the suicide
                                                   rule, reward, termination, immediate respawn of killed invaders)
            level3/components/offsets_handler.py   OffsetHandler (distances, in-range lists, last / current closest distance, dome tests)
            level3/components/quadcopter_manager.py QuadcopterManager (registry, shoot_by_ids, arm / disarm)
            core/.../weapons/gun.py                Gun
  stage01   level2/pyflyt_level2_environment_modified_v2.py   compute_reward, compute_termination, replace_invader_if_close,
                                                   update_last_distance, called in the order step() calls them (:137-146)

Run in the build container only (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_stage_logic.py

As in gen_task_logic.py these files are pure numpy on the exercised path but import pybullet / PyFlyt / gymnasium / pynput through
`Quadcopter`, the aviary simulations and the env base class; none is installable here, so STAND-IN modules are registered for exactly
those names before the reference files are loaded by path: empty packages, tripwire classes that raise on any use, and ONE data-holder
drone (`HarnessQuadcopter`, in the style of the reference's own core/entities/quadcopters/fake_quadcopter.py:7-65) that carries the
REFERENCE's Gun.  `stages.py` also asks `core.notification_system.topics_enum` for `Topics_Enum`, a name that module no longer has
(it defines `TopicsEnum`): the generator adds that alias, with the `.value` the subscription uses, before loading the file.

stage01's methods run on an instance made with `object.__new__` (its __init__ builds the PyBullet world) that is given the attributes
__init__ would have set (:35-48) and a three-method stand-in for the level2 QuadcopterManager (get_invaders / get_pursuers /
replace_invader, which records the call: the real one teleports the body and runs one PyFlyt control update, quadcopter_manager.py:166-179).
`simulation.drones` is in spawn order: invader, pursuer 0, pursuer 1 (__init__, :58-63).

Slots are the product's: stage02 pursuers first (slot p = id 10 + p), then invaders (slot P + j = id 1 + j), the reference's registry
order within each type (on_env_init spawns invaders first, stages.py:98-102); stage01 slot 0 = RL pursuer, 1 = idle pursuer, 2 = invader.
The hit draws handed to the reference's `random.random()` are the product's Philox words for (seed 0, env = arena index, RNG_HIT,
pursuer, episode 1, step): replaying arena i as env i of a te_env meets the same hits and misses.
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference/src"
OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

EPISODE, SEED, RNG_HIT = 1, 0, 3
TOUCHED = []


def tripwire(name):
    class Tripwire:
        def __init__(self, *a, **k):
            TOUCHED.append((name, "__init__"))

        def __getattr__(self, attr):
            TOUCHED.append((name, attr))
            raise AssertionError(f"stand-in {name}.{attr} was used: the exercised path is not pybullet-free")
    Tripwire.__name__ = name
    return Tripwire


def package(name):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


def module(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def by_path(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    import enum
    from core.entities.entity_type import EntityType
    from core.notification_system import topics_enum
    if not hasattr(topics_enum, "Topics_Enum"):   # stages.py:10,52 (see the header)
        topics_enum.Topics_Enum = enum.Enum("Topics_Enum", {"AGENT_STEP_BROADCAST": topics_enum.TopicsEnum.AGENT_STEP_BROADCAST})
    package("core.entities.quadcopters")
    package("core.entities.quadcopters.components")
    package("core.entities.quadcopters.components.weapons")
    gun = by_path("core.entities.quadcopters.components.weapons.gun", "core/entities/quadcopters/components/weapons/gun.py")

    class HarnessQuadcopter:
        """The public surface of Quadcopter the stage logic uses (quadcopter.py:228-229,343-366,433-478), without PyBullet."""

        def __init__(self, id_, quadcopter_type):
            self.id, self.quadcopter_type = id_, quadcopter_type
            self._armed = False
            self.gun = gun.Gun(parent_id=id_)
            self._inertial = {"position": np.zeros(3), "velocity": np.zeros(3), "attitude": np.zeros(3), "angular_rate": np.zeros(3)}
            self.replaced_to = None

        armed = property(lambda self: self._armed)
        inertial_data = property(lambda self: self._inertial)
        gun_state = property(lambda self: self.gun.get_state())

        def set_munition(self, m):
            self.gun.set_munition(m)

        def arm(self):          # quadcopter.py:445-459
            self._armed = True
            self.gun.reset()

        def disarm(self):       # quadcopter.py:461-478
            self._armed = False

        def replace(self, position, attitude=None):  # quadcopter.py:433-439
            self._inertial = dict(self._inertial, position=np.array(position, float), velocity=np.zeros(3))
            self.replaced_to = np.array(position, float)

    module("core.entities.quadcopters.quadcopter", Quadcopter=HarnessQuadcopter, EntityType=EntityType)
    # ---- stage02: level3
    for name in ("threatengage", "threatengage.environments", "threatengage.environments.level3", "threatengage.environments.level3.components",
                 "threatengage.environments.level2", "threatengage.environments.level2.components"):
        package(name)
    b3 = "threatengage.environments.level3.components"
    module(b3 + ".pyflyt_level3_simulation", L3AviarySimulation=tripwire("L3AviarySimulation"))
    qm = by_path(b3 + ".quadcopter_manager", "threatengage/environments/level3/components/quadcopter_manager.py")
    by_path(b3 + ".task_progression", "threatengage/environments/level3/components/task_progression.py")
    by_path(b3 + ".offsets_handler", "threatengage/environments/level3/components/offsets_handler.py")
    stages = by_path(b3 + ".stages", "threatengage/environments/level3/components/stages.py")
    # ---- stage01: level2
    package("pynput")
    module("pynput.keyboard", Key=tripwire("Key"), KeyCode=tripwire("KeyCode"))
    module("gymnasium", spaces=tripwire("spaces")(), Env=type("Env", (), {}))
    TOUCHED.clear()                                   # the spaces stand-in's own constructor
    b2 = "threatengage.environments.level2.components"
    module(b2 + ".pyflyt_level2_simulation", L2AviarySimulation=tripwire("L2AviarySimulation"))
    module(b2 + ".quadcopter_manager", QuadcopterManager=tripwire("QuadcopterManager"))
    by_path(b2 + ".normalization", "threatengage/environments/level2/components/normalization.py")
    env2 = by_path("threatengage.environments.level2.pyflyt_level2_environment_modified_v2",
                   "threatengage/environments/level2/pyflyt_level2_environment_modified_v2.py")
    return EntityType, gun, HarnessQuadcopter, qm, stages, env2


def philox_u01(arena, pursuer, step):
    from oracle import te_oracle as O
    r = O.philox([arena, RNG_HIT | (pursuer << 8), EPISODE, step], [SEED, 0])
    return float(int(r[0]) >> 8) / 16777216.0


class Draw:  # stands in for the `random` module inside gun.py
    queue = []

    @classmethod
    def random(cls):
        return cls.queue.pop(0)


# ======================================================================================================================= stage02
P2, I2, DOME2 = 2, 8, 8.0     # BASELINE config 2: 8 invaders; dome 8 (pyflyt_level3_environment_v2.py:32)


def stage02_arenas(rng, n):
    D = P2 + I2
    A = []

    def blank():
        a = dict(pos=np.zeros((D, 3)), prev=None, vel=np.zeros(3), munition=np.array([4, 10], np.int32), last_fired=np.array([-60, -60], np.int32),
                 step=10, armed1=1)
        a["pos"][0] = [0.5, 1.0, 1.2]; a["pos"][1] = [-1.0, 0.3, 1.5]
        for j in range(I2):
            a["pos"][P2 + j] = [2.5 + 0.3 * j, -2.0 + 0.5 * j, 2.0 + 0.2 * j]
        return a

    def case(**kw):
        a = blank()
        for j, p in kw.pop("inv", {}).items():
            a["pos"][P2 + j] = p
        for j, p in kw.pop("prev_inv", {}).items():
            if a["prev"] is None: a["prev"] = a["pos"].copy()
            a["prev"][P2 + j] = p
        for k, v in kw.items():
            if k == "p0": a["pos"][0] = v
            elif k == "p1": a["pos"][1] = v
            elif k == "prev_p0":
                if a["prev"] is None: a["prev"] = a["pos"].copy()
                a["prev"][0] = v
            else: a[k] = np.array(v) if isinstance(a[k], np.ndarray) else v
        A.append(a)

    e = lambda v: np.array(v, float)
    p0, p1 = e([0.5, 1.0, 1.2]), e([-1.0, 0.3, 1.5])
    case()                                                                         # nothing in range
    case(inv={0: p0 + e([0.5, 0, 0])})                                             # agent shoots (hit or miss by draw)
    case(inv={0: p1 + e([0, 0.6, 0])})                                             # the supporter shoots too (stages.py:205-219)
    case(inv={0: p0 + e([0.5, 0, 0])}, last_fired=[0, -60], step=30)               # cooling down: no shot; reward = d (2 reload - 1)
    case(inv={0: p0 + e([0.5, 0, 0])}, last_fired=[0, -60], step=60)               # cooldown exactly over
    case(inv={0: p0 + e([0.5, 0, 0])}, munition=[0, 10])                           # munition 0: shoot_by_ids' suicide rule = a kill without a draw
    case(inv={0: p0 + e([0.1, 0, 0])})                                             # shot AND explosion on the stale matrix
    case(inv={0: p0 + e([0.1, 0, 0])}, munition=[0, 10])
    case(inv={0: p1 + e([0.1, 0, 0])})                                             # the supporter explodes: terminal (armed pursuers < 2)
    case(p1=p0 + e([0.6, 0, 0]), inv={0: p0 + e([0.3, 0, 0])})                     # one invader in both shoot ranges (double credit)
    case(inv={0: p0 + e([0.5, 0, 0]), 1: p0 + e([0.3, 0, 0]), 2: p0 + e([0.8, 0, 0])})   # closest of three
    case(inv={0: e([0, 0, 8.3])})                                                  # invader outside the dome
    case(p0=e([0, 8.2, 1]))                                                        # agent outside the dome: -1000 and terminal
    case(p1=e([0, -8.2, 1]))                                                       # supporter outside the dome
    case(step=601); case(step=600)                                                 # time
    case(prev_p0=p0 + e([-0.5, 0.3, 0]), vel=[0.3, -0.4, 0.1])                     # approach bonus (closer by > 0.01)
    case(prev_p0=p0 + e([-0.5, 0.3, 0]), vel=[0.3, -0.4, 0.1], last_fired=[5, -60], step=20)   # ... not while reloading
    case(prev_p0=p0 + e([-0.5, 0.3, 0]), vel=[0.3, -0.4, 0.1], munition=[0, 10], last_fired=[5, -60], step=20)  # ... but with no munition
    case(prev_p0=p0 + e([0.004, 0, 0]), vel=[0.3, -0.4, 0.1])                      # closer by less than the 0.01 threshold
    case(armed1=0)                                                                 # supporter already dead: terminal
    n_scripted = len(A)
    while len(A) < n:
        a = blank()
        a["pos"][0] = rng.uniform(-2, 2, 3) * [1, 1, 0.5] + [0, 0, 1.5]
        a["pos"][1] = rng.uniform(-2, 2, 3) * [1, 1, 0.5] + [0, 0, 1.5]
        if rng.rand() < 0.15: a["pos"][1] = a["pos"][0] + rng.uniform(-0.5, 0.5, 3)
        if rng.rand() < 0.04: a["armed1"] = 0
        if rng.rand() < 0.05: a["pos"][0] *= 8.2 / np.linalg.norm(a["pos"][0])
        if rng.rand() < 0.03: a["pos"][1] *= 8.2 / np.linalg.norm(a["pos"][1])
        for j in range(I2):
            u = rng.rand()
            anchor = a["pos"][rng.randint(0, P2)]
            dirn = rng.normal(size=3); dirn /= np.linalg.norm(dirn)
            if u < 0.12: a["pos"][P2 + j] = anchor + dirn * rng.uniform(0.22, 0.95)
            elif u < 0.17: a["pos"][P2 + j] = anchor + dirn * rng.uniform(0.02, 0.18)
            elif u < 0.19: a["pos"][P2 + j] = dirn * rng.uniform(8.05, 9)
            else: a["pos"][P2 + j] = stage02_sample(rng)
        a["prev"] = a["pos"] + rng.normal(size=(D, 3)) * 0.05 * (rng.rand() < 0.8)
        a["munition"] = np.array([rng.choice([0, 1, 4]), rng.choice([0, 3, 10])], np.int32)
        a["step"] = int(rng.choice([5, 61, 150, 599, 600, 601]))
        a["last_fired"] = np.array([rng.choice([-60, a["step"] - 3, a["step"] - 60, a["step"] - 75]),
                                    rng.choice([-60, a["step"] - 10, a["step"] - 61])], np.int32)
        a["vel"] = rng.uniform(-1, 1, 3)
        ok = True   # keep every decision 1e-3 away from its threshold so that a float32 replay takes the same branch
        for p in range(P2):
            for j in range(I2):
                d = np.linalg.norm(a["pos"][p] - a["pos"][P2 + j])
                ok &= abs(d - 1.0) > 1e-3 and abs(d - 0.2) > 1e-3
        for s in range(D):
            ok &= abs(np.linalg.norm(a["pos"][s]) - DOME2) > 1e-3
        cur = min(np.linalg.norm(a["pos"][0] - a["pos"][P2 + j]) for j in range(I2))
        last = min(np.linalg.norm(a["prev"][0] - a["prev"][P2 + j]) for j in range(I2))
        ok &= abs(last - cur - 0.01) > 1e-3
        if ok:
            A.append(a)
    return A, n_scripted


def stage02_sample(rng):   # the shape of stages.py:351-372 with r in [2, 6]
    r, th, ph = rng.uniform(2, 6), rng.uniform(0, 2 * np.pi), rng.uniform(0, np.pi / 2)
    return np.array([r * np.sin(ph) * np.cos(th), r * np.sin(ph) * np.sin(th), r * np.cos(ph)])


def stage02(EntityType, gun_mod, HQ, qm_mod, stages_mod, n=224):
    from core.notification_system.message_hub import MessageHub
    from core.notification_system.topics_enum import TopicsEnum
    D = P2 + I2
    slot_of = lambda id_: id_ - 10 if id_ >= 10 else P2 + id_ - 1
    rng = np.random.RandomState(20261005)
    arenas, n_scripted = stage02_arenas(rng, n)
    keys = ("pos", "prev", "vel", "munition", "last_fired", "step", "armed", "draws", "last_min", "cur_min", "in_shoot", "in_explode", "counts",
            "reward", "done", "armed_mid", "respawned", "respawn_pos", "munition_after", "last_fired_after", "armed_after", "last_min_after", "gun_state")
    rec = {k: [] for k in keys}
    for ai, a in enumerate(arenas):
        hub = MessageHub(); hub._initialize()
        sim = types.SimpleNamespace(active_drones={})
        mgr = qm_mod.QuadcopterManager(sim)
        drones = {}
        for j in range(I2):
            drones[P2 + j] = HQ(1 + j, EntityType.LOITERINGMUNITION)
        for p in range(P2):
            drones[p] = HQ(10 + p, EntityType.LOYALWINGMAN)
        for s in list(range(P2, D)) + list(range(P2)):
            mgr.drone_registry[drones[s].id] = drones[s]
        stage = stages_mod.L3Stage1(mgr, DOME2)
        prev = a["pos"] if a["prev"] is None else a["prev"]
        mgr.arm_all()
        drones[0].gun.set_munition(4)                                  # on_episode_start (stages.py:118)
        if not a["armed1"]:
            mgr.disarm_by_quadcopter(drones[1])
        for s in range(D):
            drones[s]._inertial["position"] = np.array(prev[s], float)
        stage.offset_handler.on_episode_start()                        # last = current = the previous step's matrix
        for s in range(D):
            drones[s]._inertial["position"] = np.array(a["pos"][s], float)
        drones[0]._inertial["velocity"] = np.array(a["vel"], float)
        for p in range(P2):
            g = drones[p].gun
            g.munition = int(a["munition"][p]); g.last_fired_step = float(a["last_fired"][p])
        step = int(a["step"])
        hub.publish(topic=TopicsEnum.AGENT_STEP_BROADCAST, message={"step": step, "timestep": 1 / 15},
                    message_context=hub.create_message_context(publisher_id=0, step=step))
        assert stage.current_step == step   # subscribed under Topics_Enum.AGENT_STEP_BROADCAST.value, which equals the TopicsEnum member the hub keys on
        assert drones[0].gun.current_step == step
        draws = [philox_u01(ai, p, step) for p in range(P2)]
        oh = stage.offset_handler
        oh.on_middle_step()
        last_min, cur_min = float(oh.last_closest_pursuer_to_invader_distance), float(oh.current_closest_pursuer_to_invader_distance)

        def rng_lists(r):
            out = np.full((P2, I2), -1, np.int32)
            for pid, ids in oh.identify_invaders_in_range(r).items():
                out[slot_of(pid), :len(ids)] = [slot_of(i) for i in ids]
            return out
        in_shoot, in_explode = rng_lists(stage.PURSUER_SHOOT_RANGE), rng_lists(stage.INVADER_EXPLOSION_RANGE)
        armed_in = np.array([int(drones[s].armed) for s in range(D)], np.int32)
        Draw.queue = [draws[p] for p in range(P2) if armed_in[p] and in_shoot[p, 0] >= 0 and drones[p].gun.can_fire()]
        seen = {}
        orig = stage.compute_reward

        def spy(shots, exploded, gs):
            seen["c"], seen["gs"] = (shots, exploded), np.array(gs, float)
            seen["armed_mid"] = np.array([int(drones[s].armed) for s in range(D)], np.int32)
            return orig(shots, exploded, gs)
        stage.compute_reward = spy
        np.random.seed(ai)
        reward, done = stage.on_step_middle()
        assert not Draw.queue
        stage.on_step_end()
        respawned = np.array([int(drones[s].replaced_to is not None) for s in range(D)], np.int32)
        rpos = np.array([drones[s].replaced_to if drones[s].replaced_to is not None else np.zeros(3) for s in range(D)])
        for k, v in dict(pos=a["pos"], prev=prev, vel=a["vel"], munition=a["munition"], last_fired=a["last_fired"], step=step, armed=armed_in, draws=draws,
                         last_min=last_min, cur_min=cur_min, in_shoot=in_shoot, in_explode=in_explode, counts=np.array(seen["c"], np.int32),
                         reward=float(reward), done=int(bool(done)), armed_mid=seen["armed_mid"], respawned=respawned, respawn_pos=rpos,
                         munition_after=[drones[p].gun.munition for p in range(P2)], last_fired_after=[int(drones[p].gun.last_fired_step) for p in range(P2)],
                         armed_after=[int(drones[s].armed) for s in range(D)],
                         last_min_after=float(oh.last_closest_pursuer_to_invader_distance), gun_state=seen["gs"]).items():
            rec[k].append(v)
    out = {k: np.array(v) for k, v in rec.items()}
    c = out["counts"]
    print(f"stage02: {len(arenas)} arenas ({n_scripted} scripted); shots {int((c[:, 0] > 0).sum())} (double {int((c[:, 0] > 1).sum())}), explosions {int((c[:, 1] > 0).sum())}, "
          f"done {int(out['done'].sum())}, respawns {int(out['respawned'].sum())}, approach bonus eligible {int((out['last_min'] - out['cur_min'] > 0.01).sum())}")
    return dict(P=P2, I=I2, dome=DOME2, n_scripted=n_scripted, **out)


# ======================================================================================================================= stage01
DOME1 = 10.0   # pyflyt_level2_environment_modified_v2.py:29


def stage01(EntityType, HQ, env2_mod, n=160):
    Env = env2_mod.PyflytL2EnviromentModifiedV2
    rng = np.random.RandomState(20261006)
    keys = ("pos", "vel", "last_dist", "step", "reward", "done", "replaced", "last_dist_after")
    rec = {k: [] for k in keys}
    scripted = []
    e = lambda v: np.array(v, float)

    def case(p0, inv, p1=(3, 3, 3), vel=(0.2, -0.1, 0.3), last_dist=5.0, step=10):
        scripted.append(dict(pos=np.array([p0, p1, inv], float), vel=e(vel), last_dist=float(last_dist), step=int(step)))
    case([0.5, 0.2, 0.1], [-0.4, 0.3, 0.6])                                  # closing in: bonus 10 |v|
    case([0.5, 0.2, 0.1], [-0.4, 0.3, 0.6], last_dist=0.5)                   # moving away: no bonus
    case([0.5, 0.2, 0.1], [0.7, 0.2, 0.3])                                   # caught (d < 0.4): +1000 and the invader is replaced
    case([0.5, 0.2, 0.1], [0.5, 0.2, 0.1])                                   # d = 0
    case([0.5, 0.2, 0.1], [0, 0, 10.8])                                      # d > dome: -1000; invader outside: terminal
    case([0, 10.3, 1], [0, 9.5, 1])                                          # agent outside the dome, invader inside, d < dome
    case([0.5, 0.2, 0.1], [-0.4, 0.3, 0.6], p1=[0, 11, 0])                   # the idle pursuer outside: NOT terminal (drones[0], drones[1] only)
    case([0.5, 0.2, 0.1], [-0.4, 0.3, 0.6], step=300)
    case([0.5, 0.2, 0.1], [-0.4, 0.3, 0.6], step=301)                        # step_calls > 20 * 15
    case([6, 0, 0], [-6, 0, 0])                                              # d > dome with both inside
    n_scripted = len(scripted)
    arenas = list(scripted)
    while len(arenas) < n:
        p0 = rng.uniform(-1.5, 1.5, 3); inv = rng.uniform(-1.5, 1.5, 3); p1 = rng.uniform(-1.5, 1.5, 3)
        u = rng.rand()
        if u < 0.2: inv = p0 + rng.normal(size=3) * 0.15
        elif u < 0.28: inv = inv / np.linalg.norm(inv) * rng.uniform(10.05, 11)
        elif u < 0.36: p0 = p0 / np.linalg.norm(p0) * rng.uniform(10.05, 11)
        elif u < 0.42: p0, inv = p0 / np.linalg.norm(p0) * rng.uniform(5.5, 9), -p0 / np.linalg.norm(p0) * rng.uniform(5.5, 9)
        a = dict(pos=np.array([p0, p1, inv]), vel=rng.uniform(-1, 1, 3), last_dist=float(rng.choice([0.0, np.linalg.norm(p0 - inv) + rng.normal() * 0.2, 20.0])),
                 step=int(rng.choice([1, 17, 299, 300, 301, 302])))
        d = np.linalg.norm(p0 - inv)
        if abs(d - 0.4) > 1e-3 and abs(d - DOME1) > 1e-3 and abs(d - a["last_dist"]) > 1e-4 and all(abs(np.linalg.norm(x) - DOME1) > 1e-3 for x in (p0, inv)):
            arenas.append(a)
    for ai, a in enumerate(arenas):
        pursuers = [HQ(10, EntityType.LOYALWINGMAN), HQ(11, EntityType.LOYALWINGMAN)]
        invader = HQ(1, EntityType.LOITERINGMUNITION)
        for d, p in zip(pursuers + [invader], a["pos"]):
            d._inertial["position"] = np.array(p, float)
        pursuers[0]._inertial["velocity"] = np.array(a["vel"], float)
        calls = []

        class Manager:   # level2/components/quadcopter_manager.py: get_invaders / get_pursuers / replace_invader (:166-179)
            get_invaders = staticmethod(lambda: [invader])
            get_pursuers = staticmethod(lambda: pursuers)

            @staticmethod
            def replace_invader(drone, position, attitude):
                calls.append(np.array(position, float))
                drone.replace(position, attitude)
        env = object.__new__(Env)
        env.CATCH_DISTANCE, env.MAX_REWARD, env.dome_radius = 0.4, 1_000, DOME1            # __init__ (:35-38)
        env.rl_frequency = 15; env.max_step_calls = 20 * 15                                  # (:44-48)
        env.quadcopter_manager = Manager
        env.simulation = types.SimpleNamespace(drones=[invader, pursuers[0], pursuers[1]])   # spawn order (:58-63)
        env.step_calls = int(a["step"]); env.last_distance = float(a["last_dist"])
        np.random.seed(ai)
        reward = env.compute_reward(); done = env.compute_termination()                      # step() (:138-139)
        env.replace_invader_if_close(); env.update_last_distance()                           # (:143-144)
        for k, v in dict(pos=a["pos"], vel=a["vel"], last_dist=a["last_dist"], step=a["step"], reward=float(reward), done=int(bool(done)),
                         replaced=len(calls), last_dist_after=float(env.last_distance)).items():
            rec[k].append(v)
    out = {k: np.array(v) for k, v in rec.items()}
    print(f"stage01: {len(arenas)} arenas ({n_scripted} scripted); caught {int(out['replaced'].sum())}, done {int(out['done'].sum())}, "
          f"approach bonus {int((np.linalg.norm(out['pos'][:, 0] - out['pos'][:, 2], axis=1) < out['last_dist']).sum())}")
    return dict(dome=DOME1, n_scripted=n_scripted, **out)


def main():
    EntityType, gun_mod, HQ, qm_mod, stages_mod, env2_mod = load_reference()
    gun_mod.random = Draw
    s2 = stage02(EntityType, gun_mod, HQ, qm_mod, stages_mod)
    s1 = stage01(EntityType, HQ, env2_mod)
    assert not TOUCHED, TOUCHED
    np.savez_compressed(os.path.join(OUT, "stage_logic.npz"), seed=SEED, episode=EPISODE,
                        **{"s2_" + k: v for k, v in s2.items()}, **{"s1_" + k: v for k, v in s1.items()})
    print("tripwires touched:", sorted(set(TOUCHED)))


if __name__ == "__main__":
    main()
