#!/usr/bin/env python3
"""Golden vectors for the level5 TASK LOGIC with its six wingmen, made by RUNNING the reference's own modules:

    threatsense/level5/components/tasks_management/tasks/level5_task.py   Level5_Task: on_step_start -> on_step_middle -> on_step_end -> on_step_start
    threatsense/level5/components/entities_manager.py                     EntitiesManager (registry, agent / allies, shoot_by_ids, ...)
    core/context/offsets_handler.py                                       OffsetHandler (incl. identify_closest_ally over FIVE allies)
    core/entities/navigators/{loyalwingman_navigator,loitering_munition_navigator_air_combat_only}.py, core/.../weapons/gun.py

What exp03's fixtures (task_logic.npz, drive_logic.npz: two pursuers) cannot show: the engagement loops over six pursuers in registry
order, the reward's target chosen through the agent's CLOSEST ALLY, five behaviour trees with their own guns and formation points, the
level5 round table (12 invader slots, 8 rounds).  Same harness as gen_task_logic.py: stand-ins only for `Quadcopter`
(`HarnessQuadcopter`: a data holder with the reference's Gun, `drive()` recorded), `ImmovableStructures` (tripwire) and the package
__init__ files that import gymnasium; everything else is the reference's code.  Per arena the reference runs a whole step cycle
(gen_drive_logic.py explains the two navigator updates); the fixture keeps its inputs, the commands of step t and t+1, reward,
termination, info and the state after.  Slots: pursuer p = id 100 + p (the agent is id 100), invader j = id 1 + j.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_level5_logic.py              # level5_logic.npz       (Level5_Task)
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_level5_logic.py level5_dumb  # level5_dumb_logic.npz  (Level5DumbMultiObjectTask, all seven wingmen scripted)
"""
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_task_logic as G  # noqa: E402  (helpers only: tripwire / package / by_path / philox)

P, I = 6, 12          # level5_task.py:77-79
D = P + I
DOME = 20.0
STATES = ("WaitState", "CollideWithWingman", "CollideWithBuilding")
slot_of = lambda id_: id_ - 100 if id_ >= 100 else P + id_ - 1     # pursuer p = id 100 + p, invader j = id 1 + j
# the two task files this generator runs: (module, class, P, I, rounds, munition, armed invaders in round r, agent flown by the tree, out file)
KINDS = {
    "level5": ("level5_task", "Level5_Task", 6, 12, 8, 20, lambda r: r, False, "level5_logic.npz"),
    # Level5DumbMultiObjectTask (level5_dumb_multiobject_task.py:87-103): 6 + 1 wingmen ALL flown by the behaviour tree (:258-263), 5 invaders in round 1,
    # one more per round up to 30, munition = all the invaders of an episode, its own reward (:452-559), the agent's death does not end the episode
    "level5_dumb": ("level5_dumb_multiobject_task", "Level5DumbMultiObjectTask", 7, 30, 26, 455, lambda r: min(r - 1 + 5, 30), True, "level5_dumb_logic.npz"),
    # Level52BTEvaluationTask (level5_2bt_evaluation_task.py:82-134): two wingmen, both scripted, the same invader table, reward 0, a fixed 1 300-step
    # limit, kills counted per wingman (kills_per_drone) instead of agent / allies
    # Level5C1FusionTask (level5_c1_fusion_task.py:82-111,448-485): the agent + one scripted wingman, 10 invader slots (4 + 1 per round, 7 rounds), the
    # minimal reward whose `last_distance` is set by the first call and never again
    "level5_c1": ("level5_c1_fusion_task", "Level5C1FusionTask", 2, 10, 7, 49, lambda r: min(r - 1 + 4, 10), False, "level5_c1_logic.npz"),
    # Level5FusionTask (level5_fusion_task.py:81-112): the RL agent + 5 scripted wingmen, FIVE more invaders per round (6 rounds), the dumb task's reward,
    # the agent's death ends the episode
    "level5_fusion": ("level5_fusion_task", "Level5FusionTask", 6, 30, 6, 105, lambda r: min((r - 1) * 5 + 5, 30), False, "level5_fusion_logic.npz"),
    "level5_2bt": ("level5_2bt_evaluation_task", "Level52BTEvaluationTask", 2, 30, 26, 455, lambda r: min(r - 1 + 5, 30), True, "level5_2bt_logic.npz"),
}
KIND = "level5"


def configure(kind):
    global KIND, P, I, D
    KIND = kind
    P, I = KINDS[kind][2], KINDS[kind][3]
    D = P + I


def load_reference():
    from core.entities.entity_type import EntityType
    G.package("core.entities.quadcopters"); G.package("core.entities.quadcopters.components"); G.package("core.entities.quadcopters.components.weapons")
    gun = G.by_path("core.entities.quadcopters.components.weapons.gun", "core/entities/quadcopters/components/weapons/gun.py")

    class HarnessQuadcopter:
        """The public surface of Quadcopter the task logic and the navigators use (quadcopter.py:228-229,343-366,398-413,433-478), without PyBullet."""

        def __init__(self, id_, quadcopter_type):
            self.id, self.quadcopter_type = id_, quadcopter_type
            self._armed = False
            self.gun = gun.Gun(parent_id=id_)
            self._inertial = {"position": np.zeros(3), "velocity": np.zeros(3), "attitude": np.zeros(3), "angular_rate": np.zeros(3)}
            self.formation_position = np.zeros(3)
            self.last_drive = None
            self.quadcopter_name = f"drone{id_}"

        armed = property(lambda self: self._armed)
        inertial_data = property(lambda self: self._inertial)
        gun_state = property(lambda self: self.gun.get_state())
        is_gun_available = property(lambda self: self.gun.is_available())
        is_munition_available = property(lambda self: self.gun.has_munition())

        def drive(self, motion_command, show_name_on=False):
            self.last_drive = np.array(motion_command, float)

        def shoot(self):
            return self.gun.shoot()

        def set_munition(self, m):
            self.gun.set_munition(m)

        def set_as_agent(self):
            pass

        def arm(self):
            self._armed = True
            self.gun.reset()

        def disarm(self):
            self._armed = False

        def replace(self, position, attitude=None):
            self._inertial = dict(self._inertial, position=np.array(position, float), velocity=np.zeros(3))
            self.formation_position = np.array(position, float)

    q = types.ModuleType("core.entities.quadcopters.quadcopter")
    q.Quadcopter = HarnessQuadcopter
    sys.modules[q.__name__] = q
    G.package("core.entities.immovable_structures")
    m = types.ModuleType("core.entities.immovable_structures.immovable_structures")
    m.ImmovableStructures = G.tripwire("ImmovableStructures")
    sys.modules[m.__name__] = m
    base = "threatsense.level5.components"
    for name in ("threatsense", "threatsense.level5", base, base + ".tasks_management", base + ".tasks_management.tasks"):
        G.package(name)
    rel = "threatsense/level5/components/"
    G.by_path(base + ".normalization", rel + "normalization.py")
    G.by_path(base + ".tasks_management.task_progression", rel + "tasks_management/task_progression.py")
    em = G.by_path(base + ".entities_manager", rel + "entities_manager.py")
    mod = KINDS[KIND][0]
    task = G.by_path(base + ".tasks_management.tasks." + mod, rel + "tasks_management/tasks/" + mod + ".py")
    nav = sys.modules["core.entities.navigators.loitering_munition_navigator_air_combat_only"]
    assert sys.modules["core.entities.navigators.loyalwingman_navigator"].__file__.startswith(G.REF)
    return EntityType, gun, HarnessQuadcopter, em, task, nav


def make_arenas(rng, n):
    A = []
    while len(A) < n:
        a = dict(armed=np.zeros(D, np.int32), pos=np.zeros((D, 3)), vel=rng.uniform(-1, 1, 3), kills=rng.randint(0, 5, 3).astype(np.int32))
        a["armed"][:P] = (rng.rand(P) > 0.12).astype(np.int32)
        if rng.rand() < 0.06: a["armed"][1:P] = 0           # the agent alone: identify_closest_ally == -1
        a["armed"][0] = 1                       # the agent's death ends the episode: it is armed at the start of every step
        if KINDS[KIND][7] and rng.rand() < 0.08: a["armed"][0] = 0   # ... except where the task goes on without it
        if not a["armed"][:P].any(): a["armed"][0] = 1               # (an episode with no pursuer left has ended)
        for p in range(P):
            a["pos"][p] = rng.uniform(-3, 3, 3) * [1, 1, 0.5] + [0, 0, 1.5]
        if rng.rand() < 0.15: a["pos"][1] = a["pos"][0] + rng.uniform(-0.5, 0.5, 3)
        if rng.rand() < 0.05: a["pos"][0] *= 20.3 / np.linalg.norm(a["pos"][0])
        if rng.rand() < 0.04: a["pos"][3] *= 20.3 / np.linalg.norm(a["pos"][3])
        if rng.rand() < 0.05: a["pos"][0][2] = rng.uniform(-5.999, -4.5)
        if rng.rand() < 0.3: a["pos"][0] *= rng.uniform(4.2, 9) / np.linalg.norm(a["pos"][0])
        for j in range(I):
            a["pos"][P + j] = [3.0 + j, -2.0 + 0.5 * j, 3.0]
        n_rounds, in_round = KINDS[KIND][4], KINDS[KIND][6]
        a["round"] = int(rng.randint(1, n_rounds + 1))            # round r arms in_round(r) invaders; some are already dead
        k = rng.randint(1, min(in_round(a["round"]), 10) + 1)
        for j in rng.choice(in_round(a["round"]), k, replace=False):
            a["armed"][P + j] = 1
            u = rng.rand()
            anchor = a["pos"][rng.randint(0, P)]
            dirn = rng.normal(size=3); dirn /= np.linalg.norm(dirn)
            if u < 0.3: a["pos"][P + j] = anchor + dirn * rng.uniform(0.22, 0.95)
            elif u < 0.4: a["pos"][P + j] = anchor + dirn * rng.uniform(0.02, 0.18)
            elif u < 0.46: a["pos"][P + j] = dirn * rng.uniform(0.01, 0.18)
            elif u < 0.5: a["pos"][P + j] = dirn * rng.uniform(20.05, 22)
            else: a["pos"][P + j] = rng.uniform(-6, 6, 3) * [1, 1, 0.4] + [0, 0, 3]
        a["munition"] = rng.choice([0, 1, 5, KINDS[KIND][5]], P).astype(np.int32)
        a["step"] = int(rng.choice([5, 61, 150, 299, 300, 301, 420]))
        a["max_step"] = int(rng.choice([300, 400, 500]))
        if KIND == "level5_2bt": a["step"], a["max_step"] = int(rng.choice([5, 61, 150, 640, 1299, 1300, 1301, 1420])), 1300
        a["last_fired"] = np.array([rng.choice([-60, a["step"] - 3, a["step"] - 60, a["step"] - 75]) for _ in range(P)], np.int32)
        a["last_dist"] = float(rng.uniform(0, 10))
        a["nav"] = rng.randint(0, 3, I)
        a["formation"] = a["pos"][:P] + rng.uniform(-2, 2, (P, 3)) * (rng.rand(P, 1) < 0.8)
        ok = True   # every decision at least 1e-3 from its threshold: a float32 replay takes the same branch
        for p in range(P):
            for j in range(I):
                d = np.linalg.norm(a["pos"][p] - a["pos"][P + j])
                ok &= abs(d - 1.0) > 1e-3 and abs(d - 0.2) > 1e-3
            for p2 in range(p):                                  # closest-ally ties
                ok &= abs(np.linalg.norm(a["pos"][0] - a["pos"][p]) - np.linalg.norm(a["pos"][0] - a["pos"][p2])) > 1e-4 or p2 == 0
        for s in range(D):
            nrm = np.linalg.norm(a["pos"][s])
            ok &= abs(nrm - DOME) > 1e-3 and abs(nrm - 0.2) > 1e-3 and abs(nrm - 4.0) > 1e-3 and abs(nrm - 8.0) > 1e-3
        ok &= abs(a["pos"][0][2] + 5.0) > 1e-3 and abs(a["pos"][0][2] + 5.99) > 1e-3
        if ok:
            A.append(a)
    return A


def main(kind="level5", n=256):
    configure(kind)
    mod_name, cls_name, _, _, n_rounds, munition, in_round, agent_scripted, out_name = KINDS[kind]
    EntityType, gun_mod, HQ, em_mod, task_mod, nav_mod = load_reference()
    from core.notification_system.message_hub import MessageHub
    state_of = {"WaitState": nav_mod.WaitState, "CollideWithWingman": nav_mod.CollideWithWingmanState, "CollideWithBuilding": nav_mod.CollideWithBuildingState}

    class Draw:
        queue = []

        @classmethod
        def random(cls):
            return cls.queue.pop(0)
    gun_mod.random = Draw
    rng = np.random.RandomState(20261008 + (kind != "level5"))
    arenas = make_arenas(rng, n)
    keys = ("armed", "pos", "vel", "munition", "last_fired", "step", "max_step", "round", "last_dist", "kills", "nav", "formation", "cmd1", "nav1", "counts",
            "reward", "done", "info", "armed_mid", "armed_after", "munition_after", "last_fired_after", "max_step_after", "kills_after", "last_dist_after",
            "round_after", "shots_fired", "cmd2", "nav2", "comparable", "closest_ally", "target",
            "reset_armed", "reset_munition", "reset_last_fired", "reset_max_step", "reset_round", "reset_kills", "reset_last_dist", "reset_nav")
    rec = {k: [] for k in keys}

    def commands(drones):
        out = np.full((D, 4), np.nan)
        for s in range(D):
            if drones[s].last_drive is not None:
                out[s] = drones[s].last_drive
                drones[s].last_drive = None
        return out

    for ai, a in enumerate(arenas):
        hub = MessageHub(); hub._initialize()
        mgr = em_mod.EntitiesManager(); mgr._initialize()
        mgr.setup_simulation(types.SimpleNamespace(active_drones={}))
        drones = {}
        for j in range(I):
            drones[P + j] = HQ(1 + j, EntityType.LOITERINGMUNITION)
        for p in range(P):
            drones[p] = HQ(100 + p, EntityType.LOYALWINGMAN)
        for s in list(range(P, D)) + list(range(P)):
            mgr.drone_registry[drones[s].id] = drones[s]
        assert mgr.set_agent(drone_id=100)
        task = getattr(task_mod, cls_name)(mgr, DOME)
        assert (task.NUM_PURSUERS, getattr(task, "MAX_NUM_INVADERS", None) or task.NUM_INVADERS, task.MAX_NUMBER_OF_ROUNDS, task.MUNITION_PER_DEFENDER) == (P, I, n_rounds, munition)
        for p in range(P):
            drones[p].set_munition(task.MUNITION_PER_DEFENDER)
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
            drones[p].formation_position = np.array(a["formation"][p], float)
        for j in range(I):
            task.kamikaze_navigator.state_registry[drones[P + j].id] = state_of[STATES[a["nav"][j]]]()
        task.MAX_STEP = int(a["max_step"]); task.current_round = int(a["round"])
        task.last_closest_distance = float(a["last_dist"])
        if kind == "level5_c1":                    # compute_reward keeps `self.last_distance`: absent until the first call (arena value 0), then frozen
            if rng.rand() < 0.15: a["last_dist"] = 0.0
            else: task.last_distance = float(a["last_dist"])
        if hasattr(task, "kills_per_drone"):      # Level52BTEvaluationTask counts per wingman
            task.kills_per_drone[100]["kills"], task.kills_per_drone[101]["kills"], task.deads = (int(x) for x in a["kills"])
        else:
            task.agent_kills, task.allies_kills, task.deads = (int(x) for x in a["kills"])
        task.offset_handler.on_episode_start()
        step = int(a["step"])
        G.broadcast_step(hub, step - 1)
        task.on_step_start()                                # ---- commands of step t
        cmd1 = commands(drones)
        nav1 = np.array([STATES.index(task.kamikaze_navigator.fetch_state(drones[P + j]).name) if a["armed"][P + j] else -1 for j in range(I)], np.int32)
        G.broadcast_step(hub, step)
        assert task.current_step == step and drones[0].gun.current_step == step
        draws = [G.philox_u01(ai, p, step) for p in range(P)]
        oh = task.offset_handler
        oh.on_middle_step()
        in_shoot = {slot_of(pid): ids for pid, ids in oh.identify_invaders_in_range(task.PURSUER_SHOOT_RANGE).items()}
        fired = np.array([int(a["armed"][p] and p in in_shoot and drones[p].gun.can_fire()) for p in range(P)], np.int32)
        Draw.queue = [draws[p] for p in range(P) if fired[p]]
        ca = oh.identify_closest_ally(100)
        ca_slot = slot_of(ca) if ca != -1 else -1
        tgt = oh.identify_closest_invader(100 if ca == -1 else ca)
        seen = {}
        orig = task.compute_reward

        def spy(*args):
            seen["c"] = args
            seen["armed_mid"] = np.array([int(drones[s].armed) for s in range(D)], np.int32)
            return orig(*args)
        task.compute_reward = spy
        reward, done = task.on_step_middle()
        assert not Draw.queue
        info = task.compute_info()
        if "kills_per_drone" in info:             # the product's info row is (kills of slot 0, kills of the others, deads, wave)
            kp = info["kills_per_drone"]
            info = dict(info, agent_kills=kp[100]["kills"], allies_kills=kp[101]["kills"])
            task.agent_kills, task.allies_kills = info["agent_kills"], info["allies_kills"]
        round_before = task.current_round
        mun_after = np.array([drones[p].gun.munition for p in range(P)], np.int32)
        lf_after = np.array([int(drones[p].gun.last_fired_step) for p in range(P)], np.int32)
        if not done:
            np.random.seed(ai)
            task.on_step_end()
        armed_after = np.array([int(drones[s].armed) for s in range(D)], np.int32)
        comparable = int(not done and task.current_round == round_before)
        cmd2 = np.full((D, 4), np.nan); nav2 = np.full(I, -1, np.int32)
        if not done:
            task.on_step_start()                            # ---- commands of step t+1
            cmd2 = commands(drones)
            nav2 = np.array([STATES.index(task.kamikaze_navigator.fetch_state(drones[P + j]).name) if armed_after[P + j] else -1 for j in range(I)], np.int32)
        post = dict(max_step_after=task.MAX_STEP, kills_after=[task.agent_kills, task.allies_kills, task.deads],
                    last_dist_after=task.last_distance if kind == "level5_c1" else task.last_closest_distance, round_after=task.current_round)
        # ---- Env.reset -> Task.on_reset (on_episode_end + on_episode_start) on whatever the cycle left: the bookkeeping a reset must restore
        np.random.seed(1000 + ai)
        task.on_reset()
        if hasattr(task, "kills_per_drone"):
            rk = [task.kills_per_drone[100]["kills"], task.kills_per_drone[101]["kills"], task.deads]
        else:
            rk = [task.agent_kills, task.allies_kills, task.deads]
        reset = dict(reset_armed=[int(drones[s].armed) for s in range(D)], reset_munition=[drones[p].gun.munition for p in range(P)],
                     reset_last_fired=[int(drones[p].gun.last_fired_step) for p in range(P)], reset_max_step=task.MAX_STEP, reset_round=task.current_round,
                     reset_kills=rk, reset_last_dist=task.last_distance if kind == "level5_c1" else task.last_closest_distance,
                     reset_nav=[STATES.index(task.kamikaze_navigator.fetch_state(drones[P + j]).name) for j in range(I)])
        for k, v in dict(reset, armed=a["armed"], pos=a["pos"], vel=a["vel"], munition=a["munition"], last_fired=a["last_fired"], step=step, max_step=a["max_step"],
                         round=a["round"], last_dist=a["last_dist"], kills=a["kills"], nav=a["nav"], formation=a["formation"], cmd1=cmd1, nav1=nav1,
                         counts=np.array(seen["c"] if len(seen["c"]) else (0, 0, 0, 0, 0), np.int32), reward=float(reward), done=int(bool(done)),
                         info=[info["agent_kills"], info["allies_kills"], info["deads"], info["current_wave"]], armed_mid=seen["armed_mid"],
                         armed_after=armed_after, munition_after=mun_after, last_fired_after=lf_after, **post,
                         shots_fired=fired, cmd2=cmd2, nav2=nav2, comparable=comparable,
                         closest_ally=ca_slot, target=slot_of(tgt) if tgt != -1 else -1).items():
            rec[k].append(v)
    assert not G.TOUCHED, G.TOUCHED
    out = {k: np.array(v) for k, v in rec.items()}
    np.savez_compressed(os.path.join(G.OUT, out_name), P=P, I=I, agent_scripted=int(agent_scripted), dome=DOME, episode=G.EPISODE, seed=G.SEED, **out)
    c = out["counts"]
    print(f"{out_name}: {len(arenas)} arenas; agent shots {int((c[:, 0] > 0).sum())}, ally shots {int((c[:, 1] > 0).sum())} (two or more allies {int((c[:, 1] > 1).sum())}), "
          f"explosions {int((c[:, 2] > 0).sum())}, ally suicides {int((c[:, 3] > 0).sum())}, agent suicides {int((c[:, 4] > 0).sum())}, done {int(out['done'].sum())}, "
          f"new rounds {int((out['round_after'] != out['round']).sum())}, comparable {int(out['comparable'].sum())}, closest ally != slot 1: {int((out['closest_ally'] > 1).sum())}, "
          f"no ally {int((out['closest_ally'] < 0).sum())}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "level5")   # one task file per process (the reference's singletons and module names are per kind)
