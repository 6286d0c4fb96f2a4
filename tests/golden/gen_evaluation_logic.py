#!/usr/bin/env python3
"""Golden vectors for Evaluation_Task (level4/components/tasks_management/tasks/evaluation_task.py) with two behaviour-tree drivers, made by
RUNNING it: on_step_start -> on_step_middle -> on_step_end -> on_step_start, with the level4 OffsetHandler / EntitiesManager, both navigators
and Gun (the harness of gen_task_logic.py, the behaviour tree loaded for real; `stable_baselines3` — imported for the "nn" drivers only — is a
tripwire).  What the exp03 fixtures cannot show: EVERY pursuer obeys the tree (drive_lw, :257-275), reward 0, no invaders-in-origin rule, the
time limit only under TIME_IS_LIMITED, kills counted per wingman (lw_kills, :498-499) and the info rows of the armed wingmen (:553-574).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_evaluation_logic.py
"""
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_task_logic as G  # noqa: E402

P, I, D = G.P, G.I, G.D          # two "bt" drivers: calculate_rounds(2, 20) = 9 rounds = 9 invader slots (evaluation_task.py:96-100)
STATES = ("WaitState", "CollideWithWingman", "CollideWithBuilding")


def main(n=256):
    mods = G.load_reference(real_tree=True)
    EntityType, gun_mod, HQ, em_mod, oh_mod, _ = mods
    sb3 = types.ModuleType("stable_baselines3"); sb3.PPO = G.tripwire("PPO"); sys.modules["stable_baselines3"] = sb3
    base = "threatengage.environments.level4.components"
    ev = G.by_path(base + ".tasks_management.tasks.evaluation_task", "threatengage/environments/level4/components/tasks_management/tasks/evaluation_task.py")
    nav_mod = sys.modules["core.entities.navigators.loitering_munition_navigator_air_combat_only"]
    state_of = {"WaitState": nav_mod.WaitState, "CollideWithWingman": nav_mod.CollideWithWingmanState, "CollideWithBuilding": nav_mod.CollideWithBuildingState}
    from core.notification_system.message_hub import MessageHub

    class Draw:
        queue = []

        @classmethod
        def random(cls):
            return cls.queue.pop(0)
    gun_mod.random = Draw
    rng = np.random.RandomState(20261009)
    arenas, _ = G.make_arenas(rng, n)
    keys = ("armed", "pos", "vel", "munition", "last_fired", "step", "max_step", "round", "kills", "nav", "formation", "limited", "cmd1", "nav1", "reward", "done",
            "armed_mid", "armed_after", "munition_after", "last_fired_after", "max_step_after", "lw_kills_after", "round_after", "shots_fired", "cmd2", "nav2",
            "comparable", "info_rows")
    rec = {k: [] for k in keys}

    def commands(drones):
        out = np.full((D, 4), np.nan)
        for s in range(D):
            if drones[s].last_drive is not None:
                out[s] = drones[s].last_drive
                drones[s].last_drive = None
        return out

    for ai, a in enumerate(arenas):
        if rng.rand() < 0.1: a["armed"][0] = 0                 # the evaluation goes on without pursuer 0
        if not a["armed"][:P].any(): a["armed"][0] = 1
        limited = ai < len(arenas) // 2      # TIME_IS_LIMITED belongs to the task object: the first half of the arenas replays as one te_env, the second as another
        rng.rand()
        nav0 = rng.randint(0, 3, I)
        formation = a["pos"][:P] + rng.uniform(-2, 2, (P, 3)) * (rng.rand(P, 1) < 0.8)
        hub = MessageHub(); hub._initialize()
        mgr = em_mod.EntitiesManager(); mgr._initialize()
        mgr.setup_simulation(types.SimpleNamespace(active_drones={}))
        drones = {}
        for j in range(I):
            drones[P + j] = HQ(1 + j, EntityType.LOITERINGMUNITION)
        for p in range(P):
            drones[p] = HQ(10 + p, EntityType.LOYALWINGMAN)
            drones[p].quadcopter_name = f"bt_{p + 1}"
        for s in list(range(P, D)) + list(range(P)):
            mgr.drone_registry[drones[s].id] = drones[s]
        cfg = {"drivers": [{"type": "bt", "name": "bt_1"}, {"type": "bt", "name": "bt_2"}], "TIME_IS_LIMITED": limited, "MAX_STEP": int(a["max_step"])}
        task = ev.Evaluation_Task(mgr, G.DOME, cfg)
        assert (task.NUM_PURSUERS, task.NUM_INVADERS, task.MAX_NUMBER_OF_ROUNDS, task.munition_per_defender) == (P, I, I, 20)
        task.drivers = {f"{drones[p].id}": task.loyalwingman_navigator for p in range(P)}     # spawn_pursuer_squad (:655-663), "bt" drivers
        task.lw_kills = {f"{drones[p].id}": int(a["kills"][p]) for p in range(P)}
        for p in range(P):
            drones[p].set_munition(task.munition_per_defender)
        for s in range(D):
            d = drones[s]
            d._inertial["position"] = a["pos"][s].copy()
            d.last_drive = None
            if a["armed"][s]:
                mgr.arm_by_quadcopter(d)
        for p in range(P):
            g = drones[p].gun
            g.munition = int(a["munition"][p]); g.last_fired_step = float(a["last_fired"][p])
            drones[p].formation_position = np.array(formation[p], float)
        for j in range(I):
            task.kamikaze_navigator.state_registry[drones[P + j].id] = state_of[STATES[nav0[j]]]()
        task.current_round = int(a["round"])
        task.offset_handler.on_episode_start()
        step = int(a["step"])
        G.broadcast_step(hub, step - 1)
        task.on_step_start()
        cmd1 = commands(drones)
        nav1 = np.array([STATES.index(task.kamikaze_navigator.fetch_state(drones[P + j]).name) if a["armed"][P + j] else -1 for j in range(I)], np.int32)
        G.broadcast_step(hub, step)
        assert task.current_step == step
        draws = [G.philox_u01(ai, p, step) for p in range(P)]
        oh = task.offset_handler
        oh.on_middle_step()
        in_shoot = {G.slot_of(pid): ids for pid, ids in oh.identify_invaders_in_range(task.PURSUER_SHOOT_RANGE).items()}
        fired = np.array([int(a["armed"][p] and p in in_shoot and drones[p].gun.can_fire()) for p in range(P)], np.int32)
        Draw.queue = [draws[p] for p in range(P) if fired[p]]
        reward, done = task.on_step_middle()
        assert not Draw.queue
        armed_mid = np.array([int(drones[s].armed) for s in range(D)], np.int32)
        info = task.compute_info()
        rows = np.full((P, 5), -1, np.int64)      # (lw_kills, lw_alive, lw_munitions, current_wave, step) of the armed wingmen
        for p in range(P):
            r = info.get(f"bt_{p + 1}")
            if r is not None:
                rows[p] = [r["lw_kills"], int(r["lw_alive"]), r["lw_munitions"], r["current_wave"], r["step"]]
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
            task.on_step_start()
            cmd2 = commands(drones)
            nav2 = np.array([STATES.index(task.kamikaze_navigator.fetch_state(drones[P + j]).name) if armed_after[P + j] else -1 for j in range(I)], np.int32)
        for k, v in dict(armed=a["armed"], pos=a["pos"], vel=a["vel"], munition=a["munition"], last_fired=a["last_fired"], step=step, max_step=a["max_step"],
                         round=a["round"], kills=a["kills"], nav=nav0, formation=formation, limited=int(limited), cmd1=cmd1, nav1=nav1, reward=float(reward),
                         done=int(bool(done)), armed_mid=armed_mid, armed_after=armed_after, munition_after=mun_after, last_fired_after=lf_after,
                         max_step_after=task.MAX_STEP, lw_kills_after=[task.lw_kills[f"{drones[p].id}"] for p in range(P)], round_after=task.current_round,
                         shots_fired=fired, cmd2=cmd2, nav2=nav2, comparable=comparable, info_rows=rows).items():
            rec[k].append(v)
    bad = [t for t in G.TOUCHED if t not in G.ALLOWED]
    assert not bad, bad
    out = {k: np.array(v) for k, v in rec.items()}
    np.savez_compressed(os.path.join(G.OUT, "evaluation_logic.npz"), P=P, I=I, dome=G.DOME, episode=G.EPISODE, seed=G.SEED, agent_scripted=1, **out)
    dk = out["lw_kills_after"] - out["kills"][:, :P]
    print(f"evaluation_logic: {len(arenas)} arenas; kills by wingman 0 / 1: {int((dk[:, 0] > 0).sum())} / {int((dk[:, 1] > 0).sum())}, done {int(out['done'].sum())} "
          f"(time limit on in {int(out['limited'].sum())}), new rounds {int((out['round_after'] != out['round']).sum())}, comparable {int(out['comparable'].sum())}, "
          f"pursuer 0 dead at the start {int((out['armed'][:, 0] == 0).sum())}; tripwires touched: {sorted(set(G.TOUCHED))}")


if __name__ == "__main__":
    main()
