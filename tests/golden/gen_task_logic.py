#!/usr/bin/env python3
"""Golden vectors for the TASK LOGIC of the level4 environments, made by RUNNING the reference's own modules:

    level4/components/entities_management/offsets_handler.py   OffsetHandler (identify_*, distances)
    level4/components/entities_management/entities_manager.py  EntitiesManager (registry, shoot_by_ids, disarm_by_ids, ...)
    level4/components/tasks_management/tasks/exp03_vFinal_task.py  Exp03_vFinal_Task (on_step_middle: engagement, reward,
                                                                   termination, info; on_step_end: wave advance)
    core/entities/quadcopters/components/weapons/gun.py        Gun

Run in the build container only (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_task_logic.py

These modules are pure numpy on the exercised path, but their import lines pull pybullet / PyFlyt / gymnasium through
`Quadcopter`, `ImmovableStructures`, `LoyalWingmanBehaviorTree` and the `threatengage` package __init__.  None of those is
installable here, so the generator registers STAND-IN modules for exactly those names before loading the reference files by
path: empty packages, and tripwire classes that raise on any use (`Tripwire`).  The one drone class the logic talks to is
`HarnessQuadcopter` below: a data holder in the style of the reference's own core/entities/quadcopters/fake_quadcopter.py:7-65
(id, type, armed flag, inertial_data dict, and the REFERENCE's Gun).  The generator asserts at the end that no tripwire was
touched except the two calls listed in ALLOWED (the behaviour tree's constructor and reset(), which Exp03_vFinal_Task calls
in __init__ / advance_round and which do not touch the task state).

What is recorded per arena (one env at the moment `task.on_step_middle()` is called): the inputs (armed flags, IMU
positions, agent velocity, guns, step counters, the hit draws) and everything the reference computes from them.  Slots are
the product's: pursuers first (slot p = reference id 10 + p), then invaders (slot P + i = reference id 1 + i), which is the
reference's registry order within each type (on_env_init spawns invaders first, exp03_vFinal_task.py:248-252).

The hit draws handed to the reference's `random.random()` are the product's own Philox words for (seed 0, env = arena index,
RNG_HIT, pursuer, episode 1, step), computed with the oracle's Philox (pinned by Random123 vectors): replaying the arena as
env `arena` of a te_env therefore meets the same hit / miss outcomes with the reference's hit probability of 0.9.
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

P, I = 2, 9            # exp03_vFinal_task.py:93-103: NUM_PURSUERS = 2, NUM_INVADERS = calculate_rounds(2, 20) = 9
D = P + I
DOME = 20.0
EPISODE, SEED = 1, 0
RNG_HIT = 3            # te_device.hpp: enum { ..., RNG_HIT = 3, ... }

TOUCHED = []           # (class, attribute) of every tripwire use
ALLOWED = {("LoyalWingmanBehaviorTree", "__init__"), ("LoyalWingmanBehaviorTree", "reset")}


def tripwire(name, allow=()):
    class Tripwire:
        def __init__(self, *a, **k):
            TOUCHED.append((name, "__init__"))

        def __getattr__(self, attr):
            TOUCHED.append((name, attr))
            if attr in allow:
                return lambda *a, **k: None
            raise AssertionError(f"stand-in {name}.{attr} was used: the exercised path is not pybullet-free")
    Tripwire.__name__ = name
    return Tripwire


def package(name):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


def by_path(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference(real_tree=False):
    """real_tree: load the reference's LoyalWingmanBehaviorTree itself (pure numpy once `Quadcopter` is the stand-in) instead of its tripwire: gen_drive_logic.py"""
    from core.entities.entity_type import EntityType

    # --- stand-ins for what cannot be imported here -------------------------------------------------------------
    package("core.entities.quadcopters")
    package("core.entities.quadcopters.components")
    package("core.entities.quadcopters.components.weapons")
    gun = by_path("core.entities.quadcopters.components.weapons.gun", "core/entities/quadcopters/components/weapons/gun.py")

    class HarnessQuadcopter:
        """The public surface of Quadcopter the task logic uses (quadcopter.py:228-229,343-366,433-478), without PyBullet."""

        def __init__(self, id_, quadcopter_type):
            self.id, self.quadcopter_type = id_, quadcopter_type
            self._armed = False
            self.gun = gun.Gun(parent_id=id_)
            self._inertial = {"position": np.zeros(3), "velocity": np.zeros(3), "attitude": np.zeros(3), "angular_rate": np.zeros(3)}
            self.formation_position = np.zeros(3)

        armed = property(lambda self: self._armed)
        inertial_data = property(lambda self: self._inertial)
        gun_state = property(lambda self: self.gun.get_state())

        is_gun_available = property(lambda self: self.gun.is_available())          # quadcopter.py:356-362
        is_munition_available = property(lambda self: self.gun.has_munition())

        def drive(self, motion_command, show_name_on=False):   # quadcopter.py:398-413 (recorded; the set-point conversion is pinned elsewhere)
            self.last_drive = np.array(motion_command, float)

        def shoot(self):
            return self.gun.shoot()

        def set_munition(self, m):
            self.gun.set_munition(m)

        def arm(self):          # quadcopter.py:445-459
            self._armed = True
            self.gun.reset()

        def disarm(self):       # quadcopter.py:461-478
            self._armed = False

        def replace(self, position, attitude):  # quadcopter.py:433-439
            self._inertial = dict(self._inertial, position=np.array(position, float), velocity=np.zeros(3))
            self.formation_position = np.array(position, float)

    q = types.ModuleType("core.entities.quadcopters.quadcopter")
    q.Quadcopter = HarnessQuadcopter
    sys.modules[q.__name__] = q
    package("core.entities.immovable_structures")
    m = types.ModuleType("core.entities.immovable_structures.immovable_structures")
    m.ImmovableStructures = tripwire("ImmovableStructures")
    sys.modules[m.__name__] = m
    if real_tree:
        by_path("core.entities.navigators.loyalwingman_navigator", "core/entities/navigators/loyalwingman_navigator.py")
    else:
        m = types.ModuleType("core.entities.navigators.loyalwingman_navigator")
        m.LoyalWingmanBehaviorTree = tripwire("LoyalWingmanBehaviorTree", allow=("reset",))
        sys.modules[m.__name__] = m
    # threatengage/__init__.py imports gymnasium: empty packages instead, real modules loaded by path below
    base = "threatengage.environments.level4.components"
    for name in ("threatengage", "threatengage.environments", "threatengage.environments.level4", base, base + ".utils",
                 base + ".tasks_management", base + ".tasks_management.tasks", base + ".entities_management"):
        package(name)
    rel = "threatengage/environments/level4/components/"
    by_path(base + ".utils.normalization", rel + "utils/normalization.py")
    by_path(base + ".tasks_management.task_progression", rel + "tasks_management/task_progression.py")
    em = by_path(base + ".entities_management.entities_manager", rel + "entities_management/entities_manager.py")
    oh = by_path(base + ".entities_management.offsets_handler", rel + "entities_management/offsets_handler.py")
    task = by_path(base + ".tasks_management.tasks.exp03_vFinal_task", rel + "tasks_management/tasks/exp03_vFinal_task.py")
    return EntityType, gun, HarnessQuadcopter, em, oh, task


def philox_u01(arena, pursuer, step):
    from oracle import te_oracle as O
    r = O.philox([arena, RNG_HIT | (pursuer << 8), EPISODE, step], [SEED, 0])
    return float(int(r[0]) >> 8) / 16777216.0


def slot_of(id_):
    return id_ - 10 if id_ >= 10 else P + id_ - 1


def make_arenas(rng, n):
    """Inputs of n arenas.  The first ones are scripted corner cases; the rest are random with the interesting events made
    frequent (invaders placed inside the shoot / explosion ranges of pursuers, near the origin, beyond the dome)."""
    A = []

    def blank():
        a = dict(armed=np.zeros(D, np.int32), pos=np.zeros((D, 3)), vel=np.zeros(3), munition=np.array([20, 20], np.int32),
                 last_fired=np.array([-60, -60], np.int32), step=10, max_step=300, round=1, last_dist=5.0,
                 kills=np.zeros(3, np.int32))
        a["armed"][:P] = 1
        a["pos"][0] = [0.5, 1.0, 1.2]; a["pos"][1] = [-1.0, 0.3, 1.5]
        for j in range(I):
            a["pos"][P + j] = [3.0 + j, -2.0 + 0.5 * j, 3.0]
        return a

    def case(**kw):
        a = blank()
        inv = kw.pop("inv", {})           # {invader index: position}
        for j, p in inv.items():
            a["armed"][P + j] = 1; a["pos"][P + j] = p
        for k, v in kw.items():
            if k == "p0": a["pos"][0] = v
            elif k == "p1": a["pos"][1] = v
            elif k == "armed1": a["armed"][1] = v
            else: a[k] = np.array(v) if isinstance(a[k], np.ndarray) else v
        a["round"] = max(a["round"], int(a["armed"][P:].sum()))
        A.append(a)

    p0, p1 = np.array([0.5, 1.0, 1.2]), np.array([-1.0, 0.3, 1.5])
    e = lambda v: np.array(v, float)
    case(inv={0: e([3, 3, 3])})                                                    # nothing in range
    case(inv={0: p0 + e([0.5, 0, 0])})                                             # agent shoots (hit or miss by draw)
    case(inv={0: p1 + e([0, 0.6, 0])})                                             # ally shoots
    case(inv={0: p0 + e([0.5, 0, 0])}, last_fired=[0, -60], step=30)               # agent cooling down: no shot
    case(inv={0: p0 + e([0.5, 0, 0])}, last_fired=[0, -60], step=60)               # cooldown exactly over
    case(inv={0: p0 + e([0.5, 0, 0])}, munition=[0, 20])                           # no munition: no shot, no explosion
    case(inv={0: p0 + e([0.1, 0, 0])})                                             # shot AND explosion on the stale matrix
    case(inv={0: p0 + e([0.1, 0, 0])}, munition=[0, 20])                           # agent suicide (+1000)
    case(inv={0: p1 + e([0.1, 0, 0])}, munition=[20, 0])                           # ally suicide (+500)
    case(inv={0: p1 + e([0.1, 0, 0])}, last_fired=[-60, 5], step=20)               # ally explodes (cooling down): -1000
    case(inv={0: p0 + e([0.1, 0, 0]), 1: p1 + e([0, 0.1, 0])}, munition=[0, 0])    # both suicide
    case(p1=p0 + e([0.6, 0, 0]), inv={0: p0 + e([0.3, 0, 0])})                     # both pursuers in range of ONE invader (double credit)
    case(p1=p0 + e([0.25, 0, 0]), inv={0: p0 + e([0.12, 0, 0])})                   # one invader inside both explosion ranges
    case(inv={0: p0 + e([0.5, 0, 0]), 1: p0 + e([0.3, 0, 0]), 2: p0 + e([0.8, 0, 0])})   # closest of three
    case(inv={0: e([0.1, 0.05, 0.1])})                                             # invader in the origin
    case(inv={0: e([0.1, 0.05, 0.1]), 1: e([4, 4, 4])})
    case(inv={0: e([0, 0, 20.5])})                                                 # invader outside the dome
    case(p0=e([0, 20.5, 1]), inv={0: e([3, 3, 3])})                                # agent outside the dome
    case(p1=e([0, -20.5, 1]), inv={0: e([3, 3, 3])})                               # ally outside the dome
    case(p0=e([0.2, 0.1, -5.5]), inv={0: e([3, 3, 3])})                            # z < -5: scaled penalty
    case(p0=e([0.2, 0.1, -5.995]), inv={0: e([3, 3, 3])})                          # z < -5.99: terminal
    case(p0=e([3.0, 3.0, 1.0]), inv={0: e([3, 3, 3])})                             # 4 < |p| < 8: C8 zone term is a bonus
    case(p0=e([7.0, 5.0, 1.0]), inv={0: e([3, 3, 3])})                             # |p| > 8: a penalty
    case(inv={0: e([3, 3, 3])}, step=301)                                          # time is up
    case(inv={0: e([3, 3, 3])}, step=300)
    case(inv={0: p0 + e([0.5, 0, 0])}, step=301, max_step=300)                     # a hit extends MAX_STEP after the test? (order)
    case(armed1=0, inv={0: e([3, 3, 3])})                                          # no ally: target = agent's closest invader
    case(armed1=0, inv={0: p0 + e([0.1, 0, 0])}, munition=[0, 20])                 # last pursuer dies
    case(inv={0: e([3, 3, 3])}, last_dist=9.0, vel=[0.3, -0.4, 0.1])               # approach bonus
    case(inv={0: e([3, 3, 3])}, last_dist=9.0, vel=[0.3, -0.4, 0.1], last_fired=[5, -60], step=20)  # ... not while reloading
    case(inv={0: e([3, 3, 3])}, last_dist=0.0)
    case(inv={j: p0 + e([0.4 + 0.05 * j, 0.1 * j, 0]) for j in range(9)}, round=9)  # last round, one shot
    case(inv={0: p0 + e([0.5, 0, 0])}, round=9)                                    # last round cleared -> all rounds over (if it hits)
    case(inv={0: p0 + e([0.5, 0, 0])}, round=3)                                    # round 3 cleared -> round 4 starts
    case(inv={0: p0 + e([0.5, 0, 0]), 1: e([3, 3, 3])}, round=2)
    n_scripted = len(A)
    while len(A) < n:
        a = blank()
        a["pos"][0] = rng.uniform(-3, 3, 3) * [1, 1, 0.5] + [0, 0, 1.5]
        a["pos"][1] = rng.uniform(-3, 3, 3) * [1, 1, 0.5] + [0, 0, 1.5]
        if rng.rand() < 0.15: a["pos"][1] = a["pos"][0] + rng.uniform(-0.5, 0.5, 3)
        if rng.rand() < 0.08: a["armed"][1] = 0
        if rng.rand() < 0.05: a["pos"][0] *= 20.3 / np.linalg.norm(a["pos"][0])
        if rng.rand() < 0.05: a["pos"][0][2] = rng.uniform(-5.999, -4.5)
        if rng.rand() < 0.3: a["pos"][0] *= rng.uniform(4.2, 9) / np.linalg.norm(a["pos"][0])
        k = rng.randint(1, I + 1)
        a["round"] = int(rng.randint(k, I + 1))
        for j in rng.choice(I, k, replace=False):
            a["armed"][P + j] = 1
            u = rng.rand()
            anchor = a["pos"][rng.randint(0, P)]
            dirn = rng.normal(size=3); dirn /= np.linalg.norm(dirn)
            if u < 0.3: a["pos"][P + j] = anchor + dirn * rng.uniform(0.22, 0.95)
            elif u < 0.42: a["pos"][P + j] = anchor + dirn * rng.uniform(0.02, 0.18)
            elif u < 0.5: a["pos"][P + j] = dirn * rng.uniform(0.01, 0.18)
            elif u < 0.55: a["pos"][P + j] = dirn * rng.uniform(20.05, 22)
            else: a["pos"][P + j] = rng.uniform(-6, 6, 3) * [1, 1, 0.4] + [0, 0, 3]
        a["munition"] = np.array([rng.choice([0, 1, 5, 20]), rng.choice([0, 2, 20])], np.int32)
        a["step"] = int(rng.choice([5, 61, 150, 299, 300, 301, 420]))
        a["max_step"] = int(rng.choice([300, 400, 500]))
        a["last_fired"] = np.array([rng.choice([-60, a["step"] - 3, a["step"] - 60, a["step"] - 75]),
                                    rng.choice([-60, a["step"] - 10, a["step"] - 61])], np.int32)
        a["last_dist"] = float(rng.uniform(0, 10))
        a["vel"] = rng.uniform(-1, 1, 3)
        a["kills"] = rng.randint(0, 5, 3).astype(np.int32)
        # keep every decision at least 1e-3 away from its threshold so that a float32 replay takes the same branch
        ok = True
        for p in range(P):
            for j in range(I):
                d = np.linalg.norm(a["pos"][p] - a["pos"][P + j])
                ok &= abs(d - 1.0) > 1e-3 and abs(d - 0.2) > 1e-3
        for s in range(D):
            nrm = np.linalg.norm(a["pos"][s])
            ok &= abs(nrm - DOME) > 1e-3 and abs(nrm - 0.2) > 1e-3 and abs(nrm - 4.0) > 1e-3
        ok &= abs(a["pos"][0][2] + 5.0) > 1e-3 and abs(a["pos"][0][2] + 5.99) > 1e-3
        if ok:
            A.append(a)
    return A, n_scripted


def setup_arena(mods, ai, a, publish=True):
    """One arena as the reference's objects at the moment the environment would call task.on_step_middle(): registry, guns, task counters,
    the step broadcast delivered, the offsets of the episode start.  Returns (hub, mgr, drones, task, step, draws)."""
    EntityType, gun_mod, HQ, em_mod, oh_mod, task_mod = mods
    from core.notification_system.message_hub import MessageHub
    from core.notification_system.topics_enum import TopicsEnum
    hub = MessageHub(); hub._initialize()                      # fresh subscriptions per arena (thread-local singleton)
    mgr = em_mod.EntitiesManager(); mgr._initialize()
    mgr.setup_simulation(types.SimpleNamespace(active_drones={}))
    drones = {}
    for j in range(I):                                         # invaders first (on_env_init order), ids 1..I
        drones[P + j] = HQ(1 + j, EntityType.LOITERINGMUNITION)
    for p in range(P):
        drones[p] = HQ(10 + p, EntityType.LOYALWINGMAN)
    for s in list(range(P, D)) + list(range(P)):
        mgr.drone_registry[drones[s].id] = drones[s]
    task = task_mod.Exp03_vFinal_Task(mgr, DOME)
    assert (task.NUM_PURSUERS, task.NUM_INVADERS, task.MAX_NUMBER_OF_ROUNDS) == (P, I, I)
    for p in range(P):
        drones[p].set_munition(task.munition_per_defender)     # spawn_pursuer_squad (exp03_vFinal_task.py:641-642)
    for s in range(D):
        d = drones[s]
        d._inertial["position"] = a["pos"][s].copy()
        if s == 0:
            d._inertial["velocity"] = np.array(a["vel"], float)
        if a["armed"][s]:
            mgr.arm_by_quadcopter(d)
    for p in range(P):
        g = drones[p].gun
        g.munition = int(a["munition"][p]); g.last_fired_step = float(a["last_fired"][p])
    task.MAX_STEP = int(a["max_step"]); task.current_round = int(a["round"])
    task.last_closest_distance = float(a["last_dist"])
    task.agent_kills, task.allies_kills, task.deads = (int(x) for x in a["kills"])
    task.offset_handler.on_episode_start()
    step = int(a["step"])
    if publish:
        broadcast_step(hub, step)
        assert task.current_step == step and drones[0].gun.current_step == step
    draws = [philox_u01(ai, p, step) for p in range(P)]
    return hub, mgr, drones, task, step, draws


def broadcast_step(hub, step):
    from core.notification_system.topics_enum import TopicsEnum
    hub.publish(topic=TopicsEnum.AGENT_STEP_BROADCAST, message={"step": step, "timestep": 1 / 15},
                message_context=hub.create_message_context(publisher_id=0, step=step))


def main(n=288):
    mods = load_reference()
    EntityType, gun_mod, HQ, em_mod, oh_mod, task_mod = mods
    from core.notification_system.message_hub import MessageHub
    from core.notification_system.topics_enum import TopicsEnum

    class Draw:  # stands in for the `random` module inside gun.py: the draw of the pursuer that is shooting
        queue = []

        @classmethod
        def random(cls):
            return cls.queue.pop(0)

    gun_mod.random = Draw
    rng = np.random.RandomState(20261004)
    arenas, n_scripted = make_arenas(rng, n)
    rec = {k: [] for k in ("armed", "pos", "vel", "munition", "last_fired", "step", "max_step", "round", "last_dist", "kills", "draws",
                           "dist", "in_shoot", "in_explode", "closest_invader", "closest_pursuer", "closest_ally", "in_origin",
                           "outside", "counts", "reward", "done", "armed_mid", "munition_after", "last_fired_after", "max_step_after",
                           "kills_after", "last_dist_after", "info", "round_after", "armed_after", "shots_fired")}
    for ai, a in enumerate(arenas):
        hub, mgr, drones, task, step, draws = setup_arena(mods, ai, a)
        # a pursuer consumes its draw only if it fires; shots are processed in pursuer order
        oh = task.offset_handler
        oh.on_middle_step()
        off = oh.current_offsets
        dist = np.full((P, I), -1.0)
        for pi, pid in enumerate(off.pursuer_ids):
            for ji, jid in enumerate(off.invader_ids):
                dist[slot_of(pid), slot_of(jid) - P] = off.distances[pi, ji]

        def rng_lists(r):
            out = np.full((P, I), -1, np.int32)
            for pid, ids in oh.identify_invaders_in_range(r).items():
                out[slot_of(pid), :len(ids)] = [slot_of(i) for i in ids]
            return out
        in_shoot, in_explode = rng_lists(task.PURSUER_SHOOT_RANGE), rng_lists(task.INVADER_EXPLOSION_RANGE)
        ci = np.array([slot_of(oh.identify_closest_invader(10 + p)) if a["armed"][p] else -1 for p in range(P)], np.int32)
        cp = np.array([slot_of(oh.identify_closest_pursuer(1 + j)) if a["armed"][P + j] else -1 for j in range(I)], np.int32)
        ca = oh.identify_closest_ally(10)
        ca = slot_of(ca) if ca != -1 else -1
        origin = np.zeros(D, np.int32); outside = np.zeros(D, np.int32)
        for i_ in oh.identify_invaders_in_origin(): origin[slot_of(i_)] = 1
        for i_ in oh.identify_pursuer_outside_dome() + oh.identify_invader_outside_dome(): outside[slot_of(i_)] = 1
        # queue the draws in the order the guns will ask for them: the pursuers that can fire at a target, in id order
        Draw.queue = [draws[p] for p in range(P) if a["armed"][p] and in_shoot[p, 0] >= 0 and drones[p].gun.can_fire()]
        fired = np.array([int(a["armed"][p] and in_shoot[p, 0] >= 0 and drones[p].gun.can_fire()) for p in range(P)], np.int32)
        # ---- the reference's step: on_step_middle recomputes the offsets itself (same state -> same matrix)
        counts_seen = {}
        orig = task.compute_reward

        def spy(*args):
            counts_seen["c"] = args
            return orig(*args)
        task.compute_reward = spy
        reward, done = task.on_step_middle()
        assert not Draw.queue
        info = task.compute_info()
        armed_mid = np.array([int(drones[s].armed) for s in range(D)], np.int32)
        mun_after = np.array([drones[p].gun.munition for p in range(P)], np.int32)
        lf_after = np.array([int(drones[p].gun.last_fired_step) for p in range(P)], np.int32)
        rec_done = bool(done)
        if not rec_done:  # the environment is reset by SB3 on a terminal step; on_step_end is unobservable then
            np.random.seed(ai)
            task.on_step_end()
        for k, v in dict(armed=a["armed"], pos=a["pos"], vel=a["vel"], munition=a["munition"], last_fired=a["last_fired"], step=step,
                         max_step=a["max_step"], round=a["round"], last_dist=a["last_dist"], kills=a["kills"], draws=draws, dist=dist,
                         in_shoot=in_shoot, in_explode=in_explode, closest_invader=ci, closest_pursuer=cp, closest_ally=ca,
                         in_origin=origin, outside=outside, counts=np.array(counts_seen["c"], np.int32), reward=float(reward),
                         done=int(rec_done), armed_mid=armed_mid, munition_after=mun_after, last_fired_after=lf_after,
                         max_step_after=task.MAX_STEP, kills_after=[task.agent_kills, task.allies_kills, task.deads],
                         last_dist_after=task.last_closest_distance,
                         info=[info["agent_kills"], info["allies_kills"], info["deads"], info["current_wave"]],
                         round_after=task.current_round, armed_after=[int(drones[s].armed) for s in range(D)],
                         shots_fired=fired).items():
            rec[k].append(v)
    bad = [t for t in TOUCHED if t not in ALLOWED]
    assert not bad, bad
    out = {k: np.array(v) for k, v in rec.items()}
    np.savez_compressed(os.path.join(OUT, "task_logic.npz"), P=P, I=I, dome=DOME, episode=EPISODE, seed=SEED, n_scripted=n_scripted, **out)
    c = out["counts"]
    print(f"task_logic: {len(arenas)} arenas ({n_scripted} scripted); agent shots {int((c[:, 0] > 0).sum())}, ally shots {int((c[:, 1] > 0).sum())}, "
          f"explosions {int((c[:, 2] > 0).sum())}, ally suicides {int((c[:, 3] > 0).sum())}, agent suicides {int((c[:, 4] > 0).sum())}, "
          f"done {int(out['done'].sum())}, new rounds {int((out['round_after'] != out['round']).sum())}, "
          f"double credit {int(((c[:, 0] > 0) & (c[:, 1] > 0)).sum())}; tripwires touched: {sorted(set(TOUCHED))}")


if __name__ == "__main__":
    main()
