#!/usr/bin/env python3
"""Golden vectors for the SPAWN SAMPLERS, made by RUNNING the reference's own functions:

  exp03   level4/.../tasks/exp03_vFinal_task.py:584-608   Exp03_vFinal_Task.generate_positions(n, r, min_z=4)
                                                          call sites: setup_round (:190, r = ENEMY_BORN_RADIUS = 6, n = round),
                                                          replace_pursuers / spawn_pursuer_squad (:620,:638, r = 2)
  stage02 level3/components/stages.py:350-368             L3Stage1.generate_positions(n, r, r_max)
                                                          call sites: replace_disarmed_invaders (:375, r in [2, 6]), replace_pursuers (:381, r = 1)
  stage01 level2/pyflyt_level2_environment_modified_v2.py:101-115,154   np.random.uniform(-1, 1, 3) per drone (reset, catch respawn)

Run in the build container only (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_spawn_samplers.py

The reference draws from numpy's unseeded global stream; the product draws Philox4x32-10 words keyed on (seed; env, purpose, slot,
episode, index).  To compare the u -> xyz MAPPING, `np.random.uniform` is replaced, for the duration of each call, by
`lo + (hi - lo) * u` (what numpy computes from a random_sample u) fed with RECORDED u: the product's own words for env e of a
te_env with seed 0, episode 1 — u = (word >> 8) / 2^24, computed with the oracle's Philox (pinned by Random123 vectors).  The reference
draws ALL of a call's thetas, then ALL its phis (stage02: radii, thetas, phis); the product keys a slot's words together: the recorded u
are handed over in the reference's call order, so position i of a call is the product's slot i.  The level4 / level3 modules are loaded
with the stand-ins of gen_task_logic.py / gen_stage_logic.py (their tripwires are never touched: generate_positions uses no `self`);
stage01's three lines are the environment's own reset() lines executed through the same patched numpy (the method itself builds
PyBullet bodies, so its draws are made by calling np.random.uniform(-1, 1, 3) in its order: invader, pursuer, extra pursuer).

Stored: u and the reference's positions (float64).  Replays: tests/test_oracle_spawn_samplers.py (oracle, both precisions, through
ote_level4_position and through reset / wave advance / respawn of an OracleEnv) and tests/test_gpu_fixtures.py (te_reset + te_step
through the C ABI, positions read back with te_get_state).
"""
import importlib.util
import os
import sys

import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

N_ENVS, SEED, EPISODE = 96, 0, 1
RNG_SPAWN_INVADER, RNG_SPAWN_PURSUER, RNG_RESPAWN = 1, 2, 6   # te_device.hpp / oracle/te_oracle.c: the purposes of the draws


def gen_module(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(OUT, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def words(env, purpose, slot, episode, index):
    """The product's Philox words of one draw as u in [0, 1): counter {env, purpose | slot << 8, episode, index}, key {seed, 0}."""
    from oracle import te_oracle as O
    r = O.philox([env, purpose | (slot << 8), episode, index], [SEED, 0])
    return [float(int(x) >> 8) / 16777216.0 for x in r]


class FedUniform:
    """np.random.uniform for the duration of one reference call: low + (high - low) * u, u taken from a queue in call order."""

    def __init__(self, us):
        self.us = list(us)
        self.calls = []

    def __call__(self, low=0.0, high=1.0, size=None):
        n = 1 if size is None else int(size)
        u = np.array([self.us.pop(0) for _ in range(n)], np.float64)
        self.calls.append((float(low), float(high), n))
        out = low + (high - low) * u
        return out if size is not None else float(out[0])


def run_fed(fn, us):
    fed = FedUniform(us)
    real = np.random.uniform
    np.random.uniform = fed
    try:
        out = fn()
    finally:
        np.random.uniform = real
    assert not fed.us, "the reference consumed fewer draws than were recorded"
    return np.asarray(out, np.float64), fed.calls


def main():
    out = {"seed": np.int64(SEED), "episode": np.int64(EPISODE), "n_envs": np.int64(N_ENVS)}
    # ------------------------------------------------------------------ exp03 (level4 family)
    G4 = gen_module("gen_task_logic")
    *_, task_mod = G4.load_reference()
    gen4 = task_mod.Exp03_vFinal_Task.generate_positions
    P, I = 2, 9
    u_p = np.zeros((N_ENVS, P, 2)); pos_p = np.zeros((N_ENVS, P, 3))
    u_i = np.zeros((N_ENVS, I, I, 2)); pos_i = np.full((N_ENVS, I, I, 3), np.nan)     # [env, round - 1, invader index, .]
    for e in range(N_ENVS):
        w = [words(e, RNG_SPAWN_PURSUER, s, EPISODE, 0) for s in range(P)]
        u_p[e] = [[w[s][0], w[s][1]] for s in range(P)]
        pos_p[e], calls = run_fed(lambda: gen4(None, P, 2), [w[s][0] for s in range(P)] + [w[s][1] for s in range(P)])
        assert calls[0][:2] == (0.0, np.pi) and calls[1][:2] == (0.0, np.pi / 2)      # r = 2 < min_z = 4: phi over the whole quarter circle
        for rnd in range(1, I + 1):
            w = [words(e, RNG_SPAWN_INVADER, P + i, EPISODE, rnd) for i in range(rnd)]
            u_i[e, rnd - 1, :rnd] = [[w[i][0], w[i][1]] for i in range(rnd)]
            pos_i[e, rnd - 1, :rnd], calls = run_fed(lambda: gen4(None, rnd, 6), [w[i][0] for i in range(rnd)] + [w[i][1] for i in range(rnd)])
            assert abs(calls[1][0] - np.arccos(4 / 6)) < 1e-15 and calls[1][1] == np.pi / 2   # r = 6 >= min_z: phi from acos(4 / 6)
    out.update(l4_pursuer_u=u_p, l4_pursuer_pos=pos_p, l4_invader_u=u_i, l4_invader_pos=pos_i,
               l4_pursuer_radius=np.float64(2.0), l4_born_radius=np.float64(6.0), l4_min_z=np.float64(4.0))
    # the in-between radii the function also serves (r < min_z, r == min_z, r > min_z), on free u: the oracle's ote_level4_position directly
    rng = np.random.RandomState(20261005)
    free_r = np.array([0.5, 2.0, 3.999, 4.0, 4.001, 6.0, 8.0, 20.0])
    free_u = rng.rand(len(free_r), 40, 2)
    free_pos = np.zeros((len(free_r), 40, 3))
    for k, r in enumerate(free_r):
        free_pos[k], _ = run_fed(lambda: gen4(None, 40, float(r)), list(free_u[k, :, 0]) + list(free_u[k, :, 1]))
    out.update(l4_free_r=free_r, l4_free_u=free_u, l4_free_pos=free_pos)
    assert not [t for t in G4.TOUCHED if t not in G4.ALLOWED], G4.TOUCHED

    # ------------------------------------------------------------------ stage02 (level3)
    G3 = gen_module("gen_stage_logic")
    _, _, _, _, stages, env2 = G3.load_reference()
    gen3 = stages.L3Stage1.generate_positions
    P2, I2 = 2, 8
    u2p = np.zeros((N_ENVS, P2, 3)); pos2p = np.zeros((N_ENVS, P2, 3))
    u2i = np.zeros((N_ENVS, I2, 3)); pos2i = np.zeros((N_ENVS, I2, 3))               # reset: every invader, tag 0
    steps = np.array([1, 7, 60, 299, 600])
    u2r = np.zeros((N_ENVS, len(steps), I2, 3)); pos2r = np.zeros((N_ENVS, len(steps), I2, 3))   # respawn inside step `steps[k]` (tag = step)
    for e in range(N_ENVS):
        w = [words(e, RNG_SPAWN_PURSUER, s, EPISODE, 0) for s in range(P2)]
        u2p[e] = [w[s][:3] for s in range(P2)]
        pos2p[e], calls = run_fed(lambda: gen3(None, P2, 1), [w[s][k] for k in range(3) for s in range(P2)])
        assert [c[:2] for c in calls] == [(1.0, 1.0), (0.0, 2 * np.pi), (0.0, np.pi / 2)]
        w = [words(e, RNG_RESPAWN, P2 + j, EPISODE, 0) for j in range(I2)]
        u2i[e] = [w[j][:3] for j in range(I2)]
        pos2i[e], calls = run_fed(lambda: gen3(None, I2, 2, 6), [w[j][k] for k in range(3) for j in range(I2)])
        assert [c[:2] for c in calls] == [(2.0, 6.0), (0.0, 2 * np.pi), (0.0, np.pi / 2)]
        for k, st in enumerate(steps):
            # the reference replaces the invaders that are disarmed at that moment in ONE call (stages.py:371-376); one invader per call
            # here, so that slot j's position depends on slot j's words alone, as in the product
            for j in range(I2):
                wj = words(e, RNG_RESPAWN, P2 + j, EPISODE, int(st))
                u2r[e, k, j] = wj[:3]
                p1, _ = run_fed(lambda: gen3(None, 1, 2, 6), wj[:3])
                pos2r[e, k, j] = p1[0]
    out.update(s2_pursuer_u=u2p, s2_pursuer_pos=pos2p, s2_invader_u=u2i, s2_invader_pos=pos2i, s2_respawn_steps=steps,
               s2_respawn_u=u2r, s2_respawn_pos=pos2r)

    # ------------------------------------------------------------------ stage01 (level2): reset() draws invader, pursuer, extra pursuer (:101-115)
    u1 = np.zeros((N_ENVS, 3, 3)); pos1 = np.zeros((N_ENVS, 3, 3))                    # product slots: 0 = RL pursuer, 1 = idle pursuer, 2 = invader
    catch_steps = np.array([3, 44, 299])
    u1c = np.zeros((N_ENVS, len(catch_steps), 3)); pos1c = np.zeros((N_ENVS, len(catch_steps), 3))
    for e in range(N_ENVS):
        w_inv = words(e, RNG_SPAWN_INVADER, 2, EPISODE, 0)[:3]
        w_p0 = words(e, RNG_SPAWN_PURSUER, 0, EPISODE, 0)[:3]
        w_p1 = words(e, RNG_SPAWN_PURSUER, 1, EPISODE, 0)[:3]

        def reset_draws():   # the three draws of PyflytL2EnviromentModifiedV2.reset, in its order
            return [np.random.uniform(-1, 1, 3), np.random.uniform(-1, 1, 3), np.random.uniform(-1, 1, 3)]
        got, calls = run_fed(reset_draws, w_inv + w_p0 + w_p1)
        assert all(c == (-1.0, 1.0, 3) for c in calls)
        u1[e] = [w_p0, w_p1, w_inv]
        pos1[e] = [got[1], got[2], got[0]]
        for k, st in enumerate(catch_steps):   # replace_invader_if_close (:154): one more U(-1, 1)^3 for the invader
            wc = words(e, RNG_RESPAWN, 2, EPISODE, int(st))[:3]
            u1c[e, k] = wc
            pos1c[e, k], _ = run_fed(lambda: np.random.uniform(-1, 1, 3), wc)
    out.update(s1_u=u1, s1_pos=pos1, s1_catch_steps=catch_steps, s1_catch_u=u1c, s1_catch_pos=pos1c)
    assert not G3.TOUCHED, G3.TOUCHED

    path = os.path.join(OUT, "spawn_samplers.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: {N_ENVS} envs; exp03 pursuers r=2, invaders r=6 for rounds 1..{I}; stage02 pursuers r=1, invaders r in [2,6] "
          f"(reset + {len(steps)} respawn steps); stage01 cube draws (reset + {len(catch_steps)} catch steps); "
          f"{os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
