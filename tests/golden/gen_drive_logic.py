#!/usr/bin/env python3
"""Golden vectors for WHO IS COMMANDED WHAT around an env.step of the level4 environments, made by RUNNING the reference's own
navigators inside the reference's own task:

    core/entities/navigators/loyalwingman_navigator.py                      LoyalWingmanBehaviorTree (the ally's behaviour tree)
    core/entities/navigators/loitering_munition_navigator_air_combat_only.py  KamikazeNavigator (every invader's state machine)
    level4/.../tasks/exp03_vFinal_task.py                                   on_step_start -> on_step_middle -> on_step_end -> on_step_start
    level4/.../entities_management/{offsets_handler,entities_manager}.py, core/.../weapons/gun.py

Same harness as gen_task_logic.py (its stand-ins, its HarnessQuadcopter, whose drive() records the motion command), with the behaviour
tree loaded for real.  Per arena the reference runs

    on_step_start()      commands of step t       (navigator update #1: offsets of the episode start = the arena's positions, guns at step t-1)
    step broadcast t, on_step_middle(), on_step_end()
    on_step_start()      commands of step t+1     (update #2: the STALE offsets of on_step_middle - dead drones still listed - guns at step t)

and the fixture keeps both command sets, the invaders' states after each update and who was still armed.  The product replays an arena
with two env.steps without physics (cfg.substeps = 0): the first launch's navigator is update #1, the second's update #2; its set-point
words after each step are compared with the reference's commands (tests/test_oracle_drive_logic.py, tests/test_gpu_fixtures.py).
Arenas whose step ends the episode or starts a new round are kept but marked not comparable for update #2 (the reference then resets /
respawns at np.random positions the product draws from Philox instead).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_drive_logic.py
"""
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_task_logic as G  # noqa: E402

P, I, D = G.P, G.I, G.D
STATES = ("WaitState", "CollideWithWingman", "CollideWithBuilding")     # = TE_NAV_WAIT / TE_NAV_COLLIDE_WINGMAN / TE_NAV_COLLIDE_BUILDING


def main(n=320):
    mods = G.load_reference(real_tree=True)
    EntityType, gun_mod, HQ, em_mod, oh_mod, task_mod = mods
    nav_mod = sys.modules["core.entities.navigators.loitering_munition_navigator_air_combat_only"]
    state_of = {"WaitState": nav_mod.WaitState, "CollideWithWingman": nav_mod.CollideWithWingmanState, "CollideWithBuilding": nav_mod.CollideWithBuildingState}

    class Draw:
        queue = []

        @classmethod
        def random(cls):
            return cls.queue.pop(0)
    gun_mod.random = Draw
    rng = np.random.RandomState(20261007)
    arenas, n_scripted = G.make_arenas(rng, n)
    keys = ("armed", "pos", "vel", "munition", "last_fired", "step", "max_step", "round", "last_dist", "kills", "nav", "formation",
            "cmd1", "nav1", "cmd2", "nav2", "armed_after", "done", "round_after", "comparable", "munition_after", "last_fired_after")
    rec = {k: [] for k in keys}

    def commands(drones):
        out = np.full((D, 4), np.nan)
        for s in range(D):
            c = getattr(drones[s], "last_drive", None)
            if c is not None:
                out[s] = c
                drones[s].last_drive = None
        return out

    for ai, a in enumerate(arenas):
        nav0 = rng.randint(0, 3, I)
        formation = a["pos"][1] + rng.uniform(-2, 2, 3) * (rng.rand() < 0.8)
        hub, mgr, drones, task, step, draws = G.setup_arena(mods, ai, a, publish=False)
        assert type(task.loyalwingman_navigator).__module__ == "core.entities.navigators.loyalwingman_navigator"
        for j in range(I):
            task.kamikaze_navigator.state_registry[drones[P + j].id] = state_of[STATES[nav0[j]]]()
        drones[1].formation_position = np.array(formation, float)
        for s in range(D):
            drones[s].last_drive = None
        G.broadcast_step(hub, step - 1)                     # the guns as the previous env.step left them
        task.on_step_start()                                # ---- commands of step t
        cmd1 = commands(drones)
        nav1 = np.array([STATES.index(task.kamikaze_navigator.fetch_state(drones[P + j]).name) if a["armed"][P + j] else -1 for j in range(I)], np.int32)
        G.broadcast_step(hub, step)
        oh = task.offset_handler
        oh.on_middle_step()
        in_shoot = {G.slot_of(pid): ids for pid, ids in oh.identify_invaders_in_range(task.PURSUER_SHOOT_RANGE).items()}
        Draw.queue = [draws[p] for p in range(P) if a["armed"][p] and p in in_shoot and drones[p].gun.can_fire()]
        reward, done = task.on_step_middle()
        assert not Draw.queue
        round_before = task.current_round
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
        for k, v in dict(armed=a["armed"], pos=a["pos"], vel=a["vel"], munition=a["munition"], last_fired=a["last_fired"], step=step, max_step=a["max_step"],
                         round=a["round"], last_dist=a["last_dist"], kills=a["kills"], nav=nav0, formation=formation, cmd1=cmd1, nav1=nav1, cmd2=cmd2, nav2=nav2,
                         armed_after=armed_after, done=int(bool(done)), round_after=task.current_round, comparable=comparable,
                         munition_after=[drones[p].gun.munition for p in range(P)], last_fired_after=[int(drones[p].gun.last_fired_step) for p in range(P)]).items():
            rec[k].append(v)
    bad = [t for t in G.TOUCHED if t not in G.ALLOWED]
    assert not bad, bad
    out = {k: np.array(v) for k, v in rec.items()}
    np.savez_compressed(os.path.join(G.OUT, "drive_logic.npz"), P=P, I=I, dome=G.DOME, episode=G.EPISODE, seed=G.SEED, n_scripted=n_scripted, **out)
    ally1 = out["cmd1"][:, 1]; ok = ~np.isnan(ally1[:, 0])
    print(f"drive_logic: {len(arenas)} arenas, comparable after the step {int(out['comparable'].sum())}; ally commanded {int(ok.sum())} times at t, "
          f"{int((~np.isnan(out['cmd2'][:, 1, 0])).sum())} at t+1; invader transitions at t {int((out['nav1'] != out['nav'])[out['nav1'] >= 0].sum())}, "
          f"at t+1 {int(((out['nav2'] != out['nav1']) & (out['nav2'] >= 0)).sum())}; tripwires touched: {sorted(set(G.TOUCHED))}")


if __name__ == "__main__":
    main()
