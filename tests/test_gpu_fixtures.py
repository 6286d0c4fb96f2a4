"""The reference-made fixtures of tests/golden/ replayed THROUGH THE C ABI on the GPU and compared with the fixture itself —
not with the oracle: te_set_state -> te_step / te_observe -> outputs + te_get_state.

  task_logic.npz     OffsetHandler + EntitiesManager + Gun + Exp03_vFinal_Task.on_step_middle / on_step_end on 288 arenas
  drive_logic.npz    LoyalWingmanBehaviorTree + KamikazeNavigator inside Exp03_vFinal_Task around one env.step (commands of step t and t+1) on 320 arenas
  level5_logic.npz   Level5_Task (six wingmen) + level5 EntitiesManager + core OffsetHandler + navigators through a whole step cycle on 256 arenas
  level5_dumb_logic.npz  the same cycle of Level5DumbMultiObjectTask: seven scripted wingmen, 30 invader slots, its own reward (te_step_students)
  level5_2bt_logic.npz   ... and of Level52BTEvaluationTask: two scripted wingmen, reward 0, fixed limit, kills per wingman (te_step + te_wingman_info)
  level5_c1_logic.npz    ... and of Level5C1FusionTask: agent + one scripted wingman, 4 -> 10 invaders, the minimal reward (te_step_stacked)
  level5_fusion_logic.npz ... and of Level5FusionTask: the RL agent + five scripted wingmen, 36 drones per env (te_step_stacked)
  evaluation_logic.npz   Evaluation_Task with two "bt" drivers through the same cycle (te_step + te_wingman_info), with and without TIME_IS_LIMITED
  stage_logic.npz    stage02: L3Stage1.on_step_middle / on_step_end + level3 OffsetHandler / QuadcopterManager / Gun on 224 arenas;
                     stage01: PyflytL2EnviromentModifiedV2 reward / termination / replace_invader_if_close on 160 arenas
  lidar_math.npz     LidarMath binning of 1 000 body-frame vectors; add_features (closer wins) on 50 feature lists
  gun.npz            Gun traces (can_fire, munition, cooldown, 3-float state)
  kamikaze.npz       KamikazeNavigator (air-combat-only and cone variants): next state + command
  normalization.npz  normalize_inertial_data
  transform_features.npz  LidarMath.transform_features + add_features(invert) on 256 snapshot pairs (te_observe_stacked)
  spawn_samplers.npz generate_positions of exp03 / stage02 and stage01's cube draws on the product's own Philox words (te_reset, wave advance, respawn)

Steps that must see exactly the fixture's positions run with cfg.substeps = 0, cfg.observe_lag = 0 (no physics: the IMU read is
the loaded state)."""
import numpy as np
import pytest

from dronechase_amd import config as K
from tests import _stage_logic as S
from tests import _task_logic as T
from tests._blob import Blob

pytestmark = pytest.mark.gpu


def _gpu(cfg):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    from dronechase_amd.batched_env import BatchedEnv
    return BatchedEnv(cfg, "cuda:0")


def _load(env, blob):
    import torch
    env.set_state(torch.from_numpy(blob.w.view(np.int32)).cuda())


def _state(env, n, D):
    return Blob(env.get_state().cpu().numpy().view(np.uint32), n, D)


def _zeros(n):
    import torch
    return torch.zeros((n, 4), dtype=torch.float32, device="cuda:0")


def test_task_logic_fixture_through_the_c_abi(golden):
    from dronechase_amd import default_config
    g = golden("task_logic.npz")
    cfg = T.config(default_config, g)
    env = _gpu(cfg)
    n = cfg.n_envs
    _load(env, T.build_blob(g, env.state_words()))
    _, _, _, reward, done, info = env.step(_zeros(n), terminal=False)
    checked = T.compare(g, reward.cpu().numpy(), done.cpu().numpy(), info.cpu().numpy(), _state(env, n, cfg.n_drones))
    assert checked == n >= 200
    env.close()


def test_drive_logic_fixture_through_the_c_abi(golden):
    """The ally's behaviour tree (prepare_slot in the engage kernel's epilogue, prepare_commands_kernel after te_set_state) and the invaders'
    state machine (top of the sub-step kernel) against the reference's navigators: two steps without physics, set-point words after each."""
    from dronechase_amd import default_config
    g = golden("drive_logic.npz")
    cfg = T.config(default_config, g)
    env = _gpu(cfg)
    n, D = cfg.n_envs, cfg.n_drones
    _load(env, T.build_blob_drive(g, env.state_words()))
    env.step(_zeros(n), terminal=False)
    c1, s1 = T.compare_commands(g, _state(env, n, D), 1)
    env.step(_zeros(n), terminal=False)
    c2, s2 = T.compare_commands(g, _state(env, n, D), 2)
    assert c1 >= 600 and s1 >= 400 and c2 >= 500 and s2 >= 400, (c1, s1, c2, s2)
    env.close()


def test_level5_logic_fixture_through_the_c_abi(golden):
    """Level5_Task with six wingmen: reward (target through the closest ally), termination, info, state, and the commands of five behaviour
    trees and twelve kamikaze state machines before and after the step; te_step_stacked without physics."""
    from dronechase_amd import default_config
    g = golden("level5_logic.npz")
    cfg = T.config5(default_config, g)
    env = _gpu(cfg)
    n, D = cfg.n_envs, cfg.n_drones
    _load(env, T.build_blob_drive(g, env.state_words()))
    out = env.step_stacked(_zeros(n), terminal=False)
    reward, done, info = (x.cpu().numpy() for x in out[-3:])
    after = _state(env, n, D)
    assert T.compare(g, reward, done, info, after) == n >= 200
    c1, s1 = T.compare_commands(g, after, 1)
    env.step_stacked(_zeros(n), terminal=False)
    c2, s2 = T.compare_commands(g, _state(env, n, D), 2)
    assert c1 >= 900 and s1 >= 250 and c2 >= 700 and s2 >= 250, (c1, s1, c2, s2)
    env.reset()
    assert T.compare_reset(g, _state(env, n, D), per_wingman_kills=False) == n
    env.close()


def test_level5_dumb_logic_fixture_through_the_c_abi(golden):
    from dronechase_amd import default_config
    g = golden("level5_dumb_logic.npz")
    cfg = T.config5_dumb(default_config, g)
    env = _gpu(cfg)
    n, D = cfg.n_envs, cfg.n_drones
    _load(env, T.build_blob_drive(g, env.state_words()))
    out = env.step_students()
    reward, done, info = (x.cpu().numpy() for x in out[-3:])
    after = _state(env, n, D)
    assert T.compare(g, reward, done, info, after) == n >= 200
    c1, s1 = T.compare_commands(g, after, 1)
    env.step_students()
    c2, s2 = T.compare_commands(g, _state(env, n, D), 2)
    assert c1 >= 1000 and s1 >= 250 and c2 >= 800 and s2 >= 250, (c1, s1, c2, s2)
    env.reset()
    assert T.compare_reset(g, _state(env, n, D), per_wingman_kills=False) == n
    env.close()


def test_level5_c1_logic_fixture_through_the_c_abi(golden):
    """Level5C1FusionTask (TE_TASK_LEVEL5_C1): the minimal reward with its once-only last_distance, 4 -> 10 invaders; te_step_stacked."""
    from dronechase_amd import default_config
    g = golden("level5_c1_logic.npz")
    cfg = T.config5_c1(default_config, g)
    env = _gpu(cfg)
    n, D = cfg.n_envs, cfg.n_drones
    _load(env, T.build_blob_drive(g, env.state_words()))
    out = env.step_stacked(_zeros(n), terminal=False)
    reward, done, info = (x.cpu().numpy() for x in out[-3:])
    after = _state(env, n, D)
    assert T.compare(g, reward, done, info, after) == n >= 200
    c1, s1 = T.compare_commands(g, after, 1)
    env.step_stacked(_zeros(n), terminal=False)
    c2, s2 = T.compare_commands(g, _state(env, n, D), 2)
    assert c1 >= 300 and s1 >= 250 and c2 >= 250 and s2 >= 200, (c1, s1, c2, s2)
    # the once-only last_distance also survives the auto-reset inside the engage kernel and te_reset
    before = _state(env, n, D)
    env.reset()
    kept = _state(env, n, D)
    assert all(kept.ef(e, "LAST_DIST")[0] == before.ef(e, "LAST_DIST")[0] for e in range(n))
    env.close()


def test_level5_fusion_logic_fixture_through_the_c_abi(golden):
    """Level5FusionTask (TE_TASK_LEVEL5_FUSION): RL agent + five scripted wingmen, 36 drones per env, five more invaders per round."""
    from dronechase_amd import default_config
    g = golden("level5_fusion_logic.npz")
    cfg = T.config5_fusion(default_config, g)
    env = _gpu(cfg)
    n, D = cfg.n_envs, cfg.n_drones
    _load(env, T.build_blob_drive(g, env.state_words()))
    out = env.step_stacked(_zeros(n), terminal=False)
    reward, done, info = (x.cpu().numpy() for x in out[-3:])
    after = _state(env, n, D)
    assert T.compare(g, reward, done, info, after) == n >= 200
    c1, s1 = T.compare_commands(g, after, 1)
    env.step_stacked(_zeros(n), terminal=False)
    c2, s2 = T.compare_commands(g, _state(env, n, D), 2)
    assert c1 >= 900 and s1 >= 250 and c2 >= 700 and s2 >= 250, (c1, s1, c2, s2)
    env.reset()
    assert T.compare_reset(g, _state(env, n, D), per_wingman_kills=False) == n
    env.close()


def test_level5_2bt_logic_fixture_through_the_c_abi(golden):
    """Level52BTEvaluationTask (TE_TASK_LEVEL5_2BT): two scripted wingmen, 30 invader slots, reward 0, fixed step limit, per-wingman kills."""
    from dronechase_amd import default_config
    g = golden("level5_2bt_logic.npz")
    cfg = T.config5_2bt(default_config, g)
    env = _gpu(cfg)
    n, D = cfg.n_envs, cfg.n_drones
    _load(env, T.build_blob_drive(g, env.state_words()))
    out = env.step(_zeros(n), terminal=False)
    reward, done, info = (x.cpu().numpy() for x in out[-3:])
    after = _state(env, n, D)
    assert T.compare(g, reward, done, info, after) == n >= 200
    rows = env.wingman_info().cpu().numpy()                  # (kills, alive, munition, wave, step) per wingman
    for e in range(n):
        assert list(rows[e, :, 0]) == list(g["kills_after"][e, :2] - g["kills"][e, :2]), e
    c1, s1 = T.compare_commands(g, after, 1)
    env.step(_zeros(n), terminal=False)
    c2, s2 = T.compare_commands(g, _state(env, n, D), 2)
    assert c1 >= 800 and s1 >= 500 and c2 >= 600 and s2 >= 500, (c1, s1, c2, s2)
    env.reset()
    assert T.compare_reset(g, _state(env, n, D), per_wingman_kills=True) == n
    env.close()


def test_evaluation_logic_fixture_through_the_c_abi(golden):
    """Evaluation_Task with two behaviour-tree drivers (cfg.evaluation): termination with and without the time limit, kills per wingman and the
    info rows through te_wingman_info, the commands of both trees and nine kamikaze state machines before and after the step."""
    from dronechase_amd import default_config
    g = golden("evaluation_logic.npz")
    total = c1 = s1 = c2 = s2 = 0
    for idx, limited in T.evaluation_groups(g):
        cfg = T.evaluation_config(default_config, g, idx, limited)
        env = _gpu(cfg)
        n, D = cfg.n_envs, cfg.n_drones
        blob, sub = T.evaluation_blob(g, idx, env.state_words())
        _load(env, blob)
        out = env.step(_zeros(n), terminal=False)
        after = _state(env, n, D)
        total += T.compare_evaluation(sub, out[-3].cpu().numpy(), out[-2].cpu().numpy(), env.wingman_info().cpu().numpy(), after)
        a, b = T.compare_commands(sub, after, 1); c1 += a; s1 += b
        env.step(_zeros(n), terminal=False)
        a, b = T.compare_commands(sub, _state(env, n, D), 2); c2 += a; s2 += b
        env.close()
    assert total == len(g["step"]) and c1 >= 700 and s1 >= 400 and c2 >= 500 and s2 >= 350, (total, c1, s1, c2, s2)


def test_stage02_logic_fixture_through_the_c_abi(golden):
    from dronechase_amd import default_config
    g = golden("stage_logic.npz")
    cfg = S.config02(default_config, g)
    env = _gpu(cfg)
    n = cfg.n_envs
    _load(env, S.build_blob02(g, env.state_words()))
    _, _, _, reward, done, _ = env.step(_zeros(n), terminal=False)
    assert S.compare02(g, reward.cpu().numpy(), done.cpu().numpy(), _state(env, n, cfg.n_drones)) == n >= 200
    env.close()


def test_stage01_logic_fixture_through_the_c_abi(golden):
    from dronechase_amd import default_config
    g = golden("stage_logic.npz")
    cfg = S.config01(default_config, g)
    env = _gpu(cfg)
    n = cfg.n_envs
    _load(env, S.build_blob01(g, env.state_words()))
    _, _, _, reward, done, _ = env.step(_zeros(n), terminal=False)
    assert S.compare01(g, reward.cpu().numpy(), done.cpu().numpy(), _state(env, n, 3)) == n >= 150
    env.close()


def _rot_zyx(rpy):
    cr, sr, cp, sp, cy, sy = np.cos(rpy[0]), np.sin(rpy[0]), np.cos(rpy[1]), np.sin(rpy[1]), np.cos(rpy[2]), np.sin(rpy[2])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]]); Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    return Rz @ Ry @ Rx


def _angle_margin(theta, phi):
    """Distance (rad) of a direction to the nearest LIDAR cell boundary (cells: pi/13 in theta, 2 pi/26 in phi)."""
    a = (theta / np.pi * 13) % 1.0
    b = ((phi + np.pi) / (2 * np.pi) * 26) % 1.0
    return min(a, 1 - a) * np.pi / 13, min(b, 1 - b) * 2 * np.pi / 26


def test_lidar_binning_fixture_through_te_observe(golden):
    """LidarMath.cartesian_to_spherical / theta_index / phi_index / normalize_distance of 1 000 body-frame vectors: each one is
    the single invader of an env whose agent sits at the origin with identity attitude; the agent's own sphere must hold exactly
    one hit, in the reference's cell, with the reference's normalised range (none when the range clips to 1.0)."""
    from dronechase_amd import default_config
    g = golden("lidar_math.npz")
    vecs, n = g["vecs"], len(g["vecs"])
    cfg = default_config("exp02", n_envs=n, n_invaders=1, motor_noise=0, auto_reset=0, lidar_radius=float(g["max_radius"]))
    env = _gpu(cfg)
    b = Blob(np.zeros(env.state_words(), np.uint32), n, 2)
    for e in range(n):
        b.place(e, 0, (0, 0, 0)); b.place(e, 1, vecs[e]); b.set_ei(e, "STEP", 3); b.refresh_snapshot(e)
    _load(env, b)
    lidar = env.observe()[0].cpu().numpy()
    checked = 0
    for e in range(n):
        sph = g["sph"][e]
        mt, mp = _angle_margin(sph[1], sph[2])
        rn = float(g["norm_dist"][e])
        hits = np.argwhere(lidar[e, 0] < 1.0)
        if rn >= 1.0:
            if sph[0] > float(g["max_radius"]) * (1 + 1e-6):   # beyond the LIDAR radius: clipped to 1.0 = invisible
                assert len(hits) == 0, e
                checked += 1
            continue
        # float32 positions: a direction within 1e-5 rad of a cell boundary (or a range of ~0, whose direction is noise) may
        # legitimately land in the neighbouring cell
        if min(mt, mp) < 1e-5 or sph[0] < 1e-6:
            continue
        assert len(hits) == 1 and tuple(hits[0]) == (int(g["th_idx"][e]), int(g["ph_idx"][e])), (e, hits, g["th_idx"][e], g["ph_idx"][e])
        t, p = hits[0]
        assert abs(lidar[e, 0, t, p] - rn) < 1e-6 and abs(lidar[e, 1, t, p] - 0.2) < 1e-7 and abs(lidar[e, 2, t, p] - 0.1) < 1e-7
        checked += 1
    assert checked > 950
    env.close()


def test_closer_wins_fixture_through_te_observe(golden):
    """LidarMath.add_features (closer wins) on the reference's 50 feature lists with forced cell collisions: every feature
    becomes a drone (LOYALWINGMAN -> a pursuer slot, LOITERINGMUNITION -> an invader slot) at the world position that the
    agent — at a random position with a random attitude in the second half of the arenas — sees at (r, theta, phi); the
    agent's own sphere must equal the reference's sphere cell for cell."""
    from dronechase_amd import default_config
    g = golden("lidar_math.npz")
    feats, nf, want = g["feats"], g["n_feats"], g["closer"]
    n, R = len(nf), float(g["max_radius"])
    P, I = 12, 11
    cfg = default_config("exp03", n_envs=n, n_pursuers=P, n_invaders=I, motor_noise=0, auto_reset=0, lidar_radius=R, dome_radius=1000.0)
    env = _gpu(cfg)
    D = P + I
    b = Blob(np.zeros(env.state_words(), np.uint32), n, D)
    rng = np.random.RandomState(1)
    risky = np.zeros(n, bool)
    for e in range(n):
        rpy = np.zeros(3) if e < n // 2 else rng.uniform([-3, -1.2, -3], [3, 1.2, 3])
        own = np.zeros(3) if e < n // 2 else rng.uniform(-5, 5, 3)
        Rm = _rot_zyx(rpy)
        for s in range(D):
            b.place(e, s, (500.0 + s, 0, 0), armed=0)
        b.place(e, 0, own); b.set_f(e, 0, "OBS_EULER", rpy)
        np_, ni_ = 1, 0
        for k in range(int(nf[e])):
            r, th, ph, flag = feats[e, k, :4]
            # a feature the reference clipped to exactly 1.0 is put slightly beyond the LIDAR radius (any such range clips to
            # 1.0 and must stay invisible); at exactly R a float32 round trip would decide by its last bit
            local = (r * R if r < 1.0 else 1.001 * R) * np.array([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)])
            if abs(flag - 0.6) < 1e-9: s = np_; np_ += 1
            else: s = P + ni_; ni_ += 1
            b.place(e, s, own + Rm @ local)
            mt, mp = _angle_margin(th, ph)
            risky[e] |= min(mt, mp) < 2e-5 or r * R < 1e-3
        # two features of one cell at ranges closer than float32 can tell apart would make the winner a coin toss
        b.set_ei(e, "STEP", 3); b.refresh_snapshot(e)
    _load(env, b)
    lidar = env.observe()[0].cpu().numpy()
    checked = 0
    for e in range(n):
        if risky[e]:
            continue
        assert np.array_equal(lidar[e, 0] < 1.0, want[e, 0] < 1.0), e                 # the same cells are hit
        np.testing.assert_allclose(lidar[e], want[e], atol=2e-6, err_msg=str(e))      # range, flag, time planes
        checked += 1
    assert checked >= 40
    assert sum(int(nf[e]) - int((want[e, 0] < 1.0).sum()) for e in range(n) if not risky[e]) >= 20   # collisions were resolved
    env.close()


def test_gun_fixture_through_the_c_abi(golden):
    """Gun traces (gun.py:56-113 run by gen_golden.py): every row (broadcast step, shoot?, draw) is one env whose agent has the
    gun state the trace had before that row and, when the row shoots, an invader 0.5 m away.  After one step without physics the
    munition, the cooldown stamp, the target's fate and the 3-float gun state of the observation must be the reference's.  The
    Bernoulli draw itself is an input of the trace: rows whose draw hit run with hit_prob 1, the others with 0."""
    from dronechase_amd import default_config
    g = golden("gun.npz")
    assert float(g["cooldown"]) == 60.0 and abs(float(g["hit_prob"]) - 0.9) < 1e-12
    total = 0
    for munition in (0, 1, 4, 20):
        steps, shoot, draws = g[f"steps_{munition}"], g[f"shoot_{munition}"], g[f"draws_{munition}"]
        hit, mun, state = g[f"hit_{munition}"], g[f"mun_{munition}"], g[f"state_{munition}"]
        n = len(steps)
        mun_before = np.concatenate([[munition], mun[:-1]])
        fired = mun < mun_before
        last_fired = np.full(n, -60, np.int64)
        lf = -60
        for i in range(n):
            last_fired[i] = lf
            if fired[i]:
                lf = int(steps[i])
        for want_hit in (1, 0):
            rows = np.flatnonzero((draws < 0.9) == bool(want_hit))
            cfg = default_config("exp02", n_envs=len(rows), n_invaders=1, munition=munition, hit_prob=float(want_hit), substeps=0, observe_lag=0,
                                 motor_noise=0, auto_reset=0, n_rounds=1)
            env = _gpu(cfg)
            b = Blob(np.zeros(env.state_words(), np.uint32), len(rows), 2)
            for e, i in enumerate(rows):
                b.place(e, 0, (0, 0, 3)); b.place(e, 1, (0.5, 0, 3) if shoot[i] else (3, 3, 3))
                b.set_i(e, 0, "MUNITION", int(mun_before[i])); b.set_i(e, 0, "LAST_FIRED", int(last_fired[i]))
                b.set_ei(e, "STEP", int(steps[i]) - 1); b.set_ei(e, "MAX_STEP", 10 ** 6); b.set_ei(e, "ROUND", 1); b.set_ei(e, "EPISODE", 1)
                b.refresh_snapshot(e)
            _load(env, b)
            _, inertial, _, _, _, info = env.step(_zeros(len(rows)), terminal=False)
            inertial, info = inertial.cpu().numpy(), info.cpu().numpy()
            after = _state(env, len(rows), 2)
            for e, i in enumerate(rows):
                assert after.i(e, 0, "MUNITION") == mun[i], (munition, i)
                assert after.i(e, 0, "LAST_FIRED") == (steps[i] if fired[i] else last_fired[i]), (munition, i)
                assert (after.i(e, 1, "ARMED") == 0) == bool(hit[i]), (munition, i)
                assert info[e, 0] == int(hit[i])
                np.testing.assert_allclose(inertial[e, 12:15], state[i], atol=1e-6, err_msg=f"{munition} {i}")
            total += len(rows)
            env.close()
    assert total == 800


@pytest.mark.parametrize("variant", ["aco", "general"])
def test_kamikaze_fixture_through_the_c_abi(golden, variant):
    """KamikazeNavigator traces (both variants, run by gen_golden.py with stub offsets): every time step is one env with the
    trace's positions, armed mask and FSM states; the sub-step kernel's navigator must leave the reference's next state and the
    velocity set-point of the reference's drive() command."""
    from dronechase_amd import default_config
    g = golden("kamikaze.npz")
    P, I = int(g["P"]), int(g["I"])
    pos, mask, st_in, st_out, cmd = (g[f"{variant}_{k}"] for k in ("pos", "mask", "state_in", "state_out", "cmd"))
    n, D = len(mask), P + I
    import ctypes
    bp = (ctypes.c_float * 3)(*[float(x) for x in g[f"{variant}_building"]])
    cfg = default_config("exp03", n_envs=n, n_invaders=I, substeps=0, observe_lag=0, motor_noise=0, auto_reset=0,
                         kamikaze_cone_check=int(variant == "general"), invader_speed=float(g[f"{variant}_speed"]), building_position=bp,
                         shoot_range=0.0, explosion_range=0.0, origin_range=0.0, dome_radius=1000.0, max_step=10 ** 6)
    env = _gpu(cfg)
    b = Blob(np.zeros(env.state_words(), np.uint32), n, D)
    for e in range(n):
        for s in range(D):
            b.place(e, s, pos[e, s], armed=int((int(mask[e]) >> s) & 1))
            if s >= P:
                b.set_i(e, s, "NAV_STATE", int(st_in[e, s]))
        b.set_ei(e, "STEP", 5); b.set_ei(e, "MAX_STEP", 10 ** 6); b.set_ei(e, "ROUND", I); b.set_ei(e, "EPISODE", 1)
        b.refresh_snapshot(e)
    _load(env, b)
    env.step(_zeros(n), terminal=False)
    after = _state(env, n, D)
    checked = 0
    for e in range(n):
        for s in range(P, D):
            if not (int(mask[e]) >> s) & 1:
                continue
            assert after.i(e, s, "ARMED") == 1
            assert after.i(e, s, "NAV_STATE") == st_out[e, s], (e, s)
            d = cmd[e, s, :3]
            nrm = np.linalg.norm(d)
            v = cmd[e, s, 3] * d / (nrm if nrm > 0 else 1.0)
            np.testing.assert_allclose(after.f(e, s, "SETPOINT", 4), [v[0], v[1], 0.0, v[2]], atol=1e-6, err_msg=f"{e} {s}")
            checked += 1
    assert checked > 100
    env.close()


def test_normalization_fixture_through_te_observe(golden):
    """normalize_inertial_data (level4/components/utils/normalization.py) on 200 states incl. saturating ones: the agent's IMU
    words are loaded as they are and te_observe must return the reference's 12 floats."""
    from dronechase_amd import default_config
    g = golden("normalization.npz")
    n = len(g["pos"])
    cfg = default_config("exp02", n_envs=n, n_invaders=1, motor_noise=0, auto_reset=0, dome_radius=float(g["dome_radius"]), max_speed=float(g["max_speed"]))
    env = _gpu(cfg)
    b = Blob(np.zeros(env.state_words(), np.uint32), n, 2)
    for e in range(n):
        b.place(e, 0, (0, 0, 0)); b.place(e, 1, (3, 3, 3))
        b.set_f(e, 0, "OBS_POS", g["pos"][e]); b.set_f(e, 0, "OBS_VEL", g["vel"][e]); b.set_f(e, 0, "OBS_EULER", g["att"][e]); b.set_f(e, 0, "OBS_RATE", g["rate"][e])
        b.set_i(e, 0, "MUNITION", 20); b.set_i(e, 0, "LAST_FIRED", -60); b.set_ei(e, "STEP", 3); b.refresh_snapshot(e)
    _load(env, b)
    inertial = env.observe()[1].cpu().numpy()
    np.testing.assert_allclose(inertial[:, :12], g["out"], atol=1e-6)
    np.testing.assert_allclose(inertial[:, 12:15], np.tile([1.0, 0.0, 1.0], (n, 1)), atol=0)   # Gun().get_state() == [1, 0, 1]
    env.close()


def test_transform_features_fixture_through_the_c_abi():
    """tests/golden/transform_features.npz (the reference's LidarMath.transform_features / neighbor_sphere_from_new_frame on 256 snapshot pairs,
    `pybullet.rotateVector` = the generator's numpy stand-in): env i carries pair i in its snapshot ring (te_set_state) and te_observe_stacked
    must hand back the reference's re-projected sphere wherever it drew the neighbour — first hop on the GPU, compared with the FIXTURE.
    Run on both implementations of the stacked observation (stack_view_kernel and the LDS fallback)."""
    import os
    import torch
    from dronechase_amd import default_config
    from oracle import te_oracle as O          # stack_draws only: which neighbour / age / shuffle env i draws (shared Philox stream)
    from tests import _transform_fixture as TF
    fx = TF.load()
    cfg = default_config("level5_c1", n_envs=256, seed=11)
    for mode in ("regs", "lds"):
        if mode == "lds":
            os.environ["TE_STACKED"] = "lds"
        try:
            g = _gpu(cfg)
        finally:
            os.environ.pop("TE_STACKED", None)
        g.reset()
        b = TF.build_state(cfg, g.get_state().cpu().numpy().view(np.uint32), fx)
        g.set_state(torch.from_numpy(b.w.view(np.int32)).cuda())
        stacked, mask, *_ = g.observe_stacked()
        torch.cuda.synchronize()
        episodes = [b.ei(e, "EPISODE") for e in range(256)]
        info = TF.check(cfg, fx, stacked.cpu().numpy(), mask.cpu().numpy(), episodes, lambda e, ep: O.stack_draws(cfg, e, ep, TF.STEP, 0b11))
        assert info["hit_cells_compared"] > 500, (mode, info)
        g.close()


def test_spawn_sampler_fixture_through_the_c_abi(golden):
    """tests/golden/spawn_samplers.npz holds what the REFERENCE's generate_positions (exp03 r = 2 and r = 6, stage02 r = 1
    and r in [2, 6]) and stage01's U(-1, 1)^3 draws make of the product's own Philox words; te_reset, the wave advance and the respawns of
    te_step must land every drone there (positions read back with te_get_state)."""
    from dronechase_amd import default_config
    from tests import _spawn_samplers as SP
    g = golden("spawn_samplers.npz")
    eng = SP.Engine(make=_gpu, default_config=default_config, load=_load, state=_state, zeros=_zeros)
    n = int(g["n_envs"])
    assert SP.replay_exp03(g, eng) == n * (3 + sum(range(2, 10)))
    assert SP.replay_stage02(g, eng) == n * (10 + 8 * len(g["s2_respawn_steps"]))
    assert SP.replay_stage01(g, eng) == n * (3 + len(g["s1_catch_steps"]))
