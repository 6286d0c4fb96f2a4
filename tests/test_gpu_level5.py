"""level5 on the MI355X: te_step_stacked / te_observe_stacked through the C ABI against the
oracle on identical seeded inputs.  The stacked observation is a chain of binning decisions (own spheres of six
wingmen, then the re-projection of up to four of their old snapshots), so the oracle reports per env and step the
smallest angular distance of any binned feature to a LIDAR cell boundary (OracleEnv.stack_margins): below
CELL_MARGIN a float32 rounding may legitimately move a feature to the neighbouring cell.  Such envs must stay a small
fraction and every mismatch must be one of them (or an env the step-logic margins flag, tests/test_gpu_parity.py).

Tolerances: OBS_TOL 1e-5 on sphere values (r_hat in [0,1], flag, time), masks / ring stamps / feature counts exact,
RING_TOL 1e-3 on the float words of the snapshot ring (positions in metres, angles in radians) after a free-running
rollout of up to 25 steps per episode."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OBS_TOL = 1e-5
RING_TOL = 1e-3      # free-running rollout: the same drift bound as tests/test_gpu_parity.py::test_rollout_parity_from_reset (measured 8e-5)
CELL_MARGIN = 5e-5   # rad; a cell is 0.24 rad wide, float32 angle errors are ~1e-6
MARGIN = 1e-4        # step-logic thresholds (metres)


def _gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box (no CPU fallback exists)")
    return torch


def _pair(N, task="level5", **over):
    torch = _gpu()
    from dronechase_amd import default_config
    from dronechase_amd.batched_env import BatchedEnv
    from oracle import te_oracle as O
    cfg = default_config(task, n_envs=N, **over)
    return torch, cfg, BatchedEnv(cfg, "cuda:0"), O.OracleEnv(cfg, "f32")


@pytest.mark.parametrize("task", ["level5", "level5_c1", "level5_fusion"])   # Level5_Task (6 wingmen, 12 slots), Level5C1FusionTask (2, 10, its own reward), Level5FusionTask (6, 30: 36 drones per env)
def test_stacked_rollout_parity_with_resets_and_rounds(task):
    N, STEPS = 512, 60
    torch, cfg, g, o = _pair(N, task, motor_noise=0, max_step=25, seed=7)
    g.reset(); o.reset()
    gs, gm, *_ = g.observe_stacked(); os_, om, *_ = o.observe_stacked()
    assert (gs == 1).all().item() and (gm == 0).all().item() and (os_ == 1).all() and (om == 0).all()
    dirty_state = np.zeros(N, bool)       # state may have legitimately diverged (ambiguous step decision); until the env resets
    dirty_cell_until = np.full(N, -1)     # a cell flip lives in the ring for at most 9 more steps (or until the env resets)
    flagged = dones = compared = rewards_compared = 0
    for t in range(STEPS):
        a = o.random_actions(3, t)
        s, m, inert, la, r, d, info = o.step_stacked(a)
        gs, gm, gi, gl, gr, gd, ginfo = g.step_stacked(torch.from_numpy(a).cuda())
        torch.cuda.synchronize()
        gs_, gm_, gd_ = gs.cpu().numpy(), gm.cpu().numpy(), gd.cpu().numpy()
        cell_amb = o.stack_margins() < CELL_MARGIN
        dirty_state |= o.state_margins() < MARGIN
        dirty_cell_until[cell_amb] = t + 9
        done = d != 0
        bad_main = (gm_ != m).any(1) | (np.abs(gs_ - s).reshape(N, -1).max(1) > OBS_TOL) | (gd_ != d)
        bad_term = np.zeros(N, bool)
        if done.any():  # SB3 terminal observation of auto-reset envs
            bad_term = done & ((np.abs(g.t_stacked.cpu().numpy() - o.t_stacked).reshape(N, -1).max(1) > OBS_TOL)
                               | (g.t_mask.cpu().numpy() != o.t_mask).any(1))
            dones += int(done.sum())
        clean = ~(dirty_state | (dirty_cell_until >= t))
        assert not ((bad_main | bad_term) & clean).any(), (t, np.nonzero((bad_main | bad_term) & clean)[0][:8])
        rew_ok = clean & (o.margins() > 1e-3)      # reward-only thresholds (approach terms) need their own margin
        gr_ = gr.cpu().numpy()
        assert (np.abs(gr_ - r)[rew_ok] <= 2e-3 + 1e-5 * np.abs(r[rew_ok])).all(), (t, np.nonzero(rew_ok & (np.abs(gr_ - r) > 2e-3 + 1e-5 * np.abs(r)))[0][:8])
        rewards_compared += int(rew_ok.sum())
        flagged += int(cell_amb.sum()); compared += int(clean.sum())
        # an auto-reset restarts from a deterministic state with an empty ring: clean again
        dirty_state[done & (gd_ == d)] = False
        dirty_cell_until[done & (gd_ == d)] = -1
    assert dones >= N            # every env auto-reset at least once (max_step 25)
    # cell-boundary ambiguities grow with the number of features in view: 18 drones per env in level5, 36 in level5_fusion
    assert flagged < 0.05 * max(1.0, cfg.n_drones / 18) * N * STEPS and compared > 0.5 * N * STEPS and rewards_compared > 0.4 * N * STEPS, (flagged, compared, rewards_compared)
    dirty = dirty_state | (dirty_cell_until >= STEPS - 1)
    # the ring itself: stamps and feature counts exact, float words close, on envs that never met an ambiguity
    from dronechase_amd import config as K
    wg, wo = g.get_state().cpu().numpy().view(np.uint32), o.get_state()
    rg, ro = o.ring(wg), o.ring(wo)
    ok = ~dirty
    assert ok.sum() > N // 4
    rg, ro = rg[ok], ro[ok]
    np.testing.assert_array_equal(rg[..., 0], ro[..., 0])          # stamps (0 = empty; the rest of an empty entry is unspecified)
    live = ro[..., 0] != 0
    assert live.sum() > (1000 if task == "level5" else 300)
    rg, ro = rg[live], ro[live]
    np.testing.assert_array_equal(rg[:, 1], ro[:, 1])              # kept features
    pose = [2, 3, 4, 5, 6, 7, 8]
    dif = np.abs(rg[:, pose].view(np.float32).astype(np.float64) - ro[:, pose].view(np.float32))
    assert dif.max() < RING_TOL, dif.max()
    for k in range(cfg.n_drones - 1):  # feature slots below the count (the rest of an entry is unspecified)
        has = ro[:, 1] > k
        fl = [12 + 4 * k + i for i in range(3)]
        dif = np.abs(rg[has][:, fl].view(np.float32).astype(np.float64) - ro[has][:, fl].view(np.float32))
        assert dif.size == 0 or dif.max() < RING_TOL, (k, dif.max())
        np.testing.assert_array_equal(rg[has][:, 15 + 4 * k], ro[has][:, 15 + 4 * k])   # entity type | publisher slot


def test_observe_stacked_reproduces_the_step_observation_and_blob_roundtrip():
    N = 256
    torch, cfg, g, o = _pair(N, motor_noise=1, seed=5)
    g.reset(); o.reset()
    for t in range(14):
        a = o.random_actions(9, t)
        gs, gm, *_ = g.step_stacked(torch.from_numpy(a).cuda())
    step_obs, step_mask = gs.clone(), gm.clone()
    done = g.done.cpu().numpy() != 0
    gs2, gm2, *_ = g.observe_stacked()
    torch.cuda.synchronize()
    keep = ~done  # an auto-reset env now shows its reset observation instead
    assert torch.equal(gm2[keep], step_mask[keep]) and torch.equal(gs2[keep], step_obs[keep])
    # state blob (drone records, env records, ring) -> a second te_env -> identical future
    from dronechase_amd.batched_env import BatchedEnv
    h = BatchedEnv(cfg, "cuda:0")
    w = g.get_state()
    assert w.numel() == N * (18 * 58 + 16 + 6 * 10 * 80)
    h.set_state(w)
    for t in range(14, 20):
        a = torch.from_numpy(o.random_actions(9, t)).cuda()
        ra, rb = g.step_stacked(a), h.step_stacked(a)
        torch.cuda.synchronize()
        for x, y in zip(ra, rb):
            assert torch.equal(x, y)
    assert torch.equal(g.get_state(), h.get_state())


def test_stacked_full_size_properties_and_api_errors():
    """65 536 level5 envs (1.6 GB of observation per step): structural invariants without the oracle."""
    torch = _gpu()
    from dronechase_amd import _lib, default_config
    from dronechase_amd.batched_env import BatchedEnv
    N = 65536
    cfg = default_config("level5", n_envs=N, seed=1)
    g = BatchedEnv(cfg, "cuda:0")
    g.reset()
    a = torch.empty((N, 4), device="cuda:0")
    for t in range(12):
        g.random_actions(5, t, out=a)
        s, m, *_ = g.step_stacked(a, terminal=False)
    torch.cuda.synchronize()
    nv = m.sum(1)
    assert int(nv.min()) >= 1 and int(nv.max()) <= 5
    hit = s[:, :, 0] < 1
    assert not bool(hit[m == 0].any())                                  # padding spheres are empty
    assert bool(((s[:, :, 1][hit] - 0.2).abs() < 1e-6).logical_or((s[:, :, 1][hit] - 0.6).abs() < 1e-6).all())
    tt = (s[:, :, 2][hit] * 10).round()
    assert bool(((s[:, :, 2][hit] * 10 - tt).abs() < 1e-5).all()) and int(tt.min()) >= 1 and int(tt.max()) <= 9
    assert abs(float((nv == 1).float().mean()) - 0.0) < 0.2             # by step 12 most drawn snapshots exist
    # determinism
    h = BatchedEnv(cfg, "cuda:0"); h.reset()
    for t in range(12):
        h.random_actions(5, t, out=a)
        s2, m2, *_ = h.step_stacked(a, terminal=False)
    torch.cuda.synchronize()
    assert torch.equal(s, s2) and torch.equal(m, m2)
    # API discipline: the two step entry points do not mix
    with pytest.raises(_lib.TEError, match="te_step_stacked"):
        g.step(a)
    plain = BatchedEnv(default_config("exp03", n_envs=64), "cuda:0")
    with pytest.raises(_lib.TEError, match="stacked_obs"):
        plain.L.te_step_stacked  # attribute exists
        _lib.check(plain.L.te_observe_stacked(plain._h, None, None, None, None, None), "te_observe_stacked")


@pytest.mark.parametrize("cls_name", ["Level5Environment", "Level5FusionEnvironment", "Level5C1FusionEnvironment"])
def test_level5_vecenv_and_single_env_surface(cls_name):
    """The reference's level5 student observation dict (level5_envrionment.py:312-351,359-362) through the SB3 VecEnv
    mirror and the single-env class; the fusion environments share it (level5_fusion_environment.py, level5_c1_fusion_environment.py:23-60)."""
    _gpu()
    from dronechase_amd import envs
    from dronechase_amd.pipeline import ReinforcementLearningPipeline
    Level5Environment = getattr(envs, cls_name)
    n = 96
    v = ReinforcementLearningPipeline.create_vectorized_environment(Level5Environment, {"rl_frequency": 15},
                                                                    n_envs=n, monitor=False, max_step=5)
    assert v.observation_space["stacked_spheres"].shape == (6, 3, 13, 26) and v.observation_space["validity_mask"].shape == (6,)
    obs = v.reset()
    assert set(obs) == {"stacked_spheres", "validity_mask", "inertial_data", "last_action"}
    assert obs["stacked_spheres"].shape == (n, 6, 3, 13, 26) and obs["validity_mask"].dtype == bool and not obs["validity_mask"].any()
    saw_terminal = False
    for t in range(8):
        a = np.tile(np.array([[0.3, -0.2, 0.1, 0.5]], np.float32), (n, 1))
        obs, rew, dones, infos = v.step(a)
        assert obs["validity_mask"].shape == (n, 6) and rew.shape == (n,)
        for i in np.flatnonzero(dones):
            t_obs = infos[i]["terminal_observation"]
            assert t_obs["stacked_spheres"].shape == (6, 3, 13, 26) and t_obs["validity_mask"].sum() >= 1
            assert not obs["validity_mask"][i].any()      # reset observation of the auto-reset env
            saw_terminal = True
        live = ~dones
        assert (obs["validity_mask"][live].sum(1) >= 1).all()
    assert saw_terminal
    v.close()
    e = Level5Environment(rl_frequency=15)
    o, info = e.reset()
    assert o["stacked_spheres"].shape == (6, 3, 13, 26) and not o["validity_mask"].any()
    o, r, term, trunc, info = e.step(np.array([0, 0, 1, 0.5], np.float32))
    assert o["validity_mask"].sum() >= 1 and trunc is False
    assert info == {} if cls_name == "Level5C1FusionEnvironment" else set(info) >= {"agent_kills", "allies_kills", "deads", "current_wave"}
    e.close()


@pytest.mark.parametrize("task", ["level5", "level5_fusion", "level5_dumb"])
def test_heterogeneous_chunk_ring_matches_lds_fallback_and_oracle(task, monkeypatch):
    """A chunk whose lanes differ: wingmen dead in scattered envs, a different highest armed slot per env, N not a multiple of 64.
    ring_push_kernel reduces the wave's highest armed slot over lanes of which some publish nothing (the round-2 butterfly ran AFTER those
    lanes had left and lost features): the default register kernels must agree with the LDS fallback (TE_STACKED=lds) cell for cell and,
    within the usual tolerances, with the oracle (tools/hetero_debug.py prints where they differ)."""
    torch = _gpu()
    from dronechase_amd import default_config
    from dronechase_amd.batched_env import BatchedEnv
    from oracle import te_oracle as O
    from tests._blob import Blob
    N = 200
    cfg = default_config(task, n_envs=N, motor_noise=0, seed=21)
    P, D = int(cfg.n_pursuers), int(cfg.n_drones)
    students = task == "level5_dumb"
    g = BatchedEnv(cfg, "cuda:0")
    monkeypatch.setenv("TE_STACKED", "lds")
    h = BatchedEnv(cfg, "cuda:0")
    monkeypatch.delenv("TE_STACKED")
    o = O.OracleEnv(cfg, "f32", threads=4)
    g.reset()
    for t in range(4):   # a few ordinary steps: the ring holds entries of every wingman
        if students: g.step_students()
        else: g.step_stacked(g.random_actions(4, t))
    b = Blob(g.get_state().cpu().numpy().view(np.uint32), N, D)
    rng = np.random.default_rng(5)
    tops = set()
    for e in range(N):
        # the agent (slot 0) lives except in a few student envs (level5_dumb survives its death); every other wingman dies with p = 0.4
        for p in range(0 if students else 1, P):
            if rng.random() < (0.15 if p == 0 else 0.4):
                b.set_i(e, p, "ARMED", 0)
                for name in ("VEL", "OMEGA"): b.set_f(e, p, name, [0, 0, 0])
        if not any(b.i(e, p, "ARMED") for p in range(P)):
            b.set_i(e, P - 1, "ARMED", 1)
        # invaders: an arbitrary subset of the table per env, so the highest armed slot differs from lane to lane
        k = int(rng.integers(1, D - P + 1))
        for d in range(P, D):
            want = (d - P) < k and rng.random() < 0.7
            if want and not b.i(e, d, "ARMED"):
                u = rng.normal(size=3); u /= np.linalg.norm(u)
                b.place(e, d, (u * rng.uniform(1.5, 5.5)).astype(np.float32)); b.hover_ready(e, d, cfg)
            elif not want and b.i(e, d, "ARMED"):
                b.set_i(e, d, "ARMED", 0)
                for name in ("VEL", "OMEGA"): b.set_f(e, d, name, [0, 0, 0])
        if not any(b.i(e, d, "ARMED") for d in range(P, D)):
            u = rng.normal(size=3); u /= np.linalg.norm(u)
            b.place(e, P, (u * 3.0).astype(np.float32)); b.hover_ready(e, P, cfg)
        b.refresh_snapshot(e)
        tops.add(b.armed_mask(e).bit_length())
    assert len(tops) >= 4                                   # the lanes of a chunk really have different highest armed slots
    w = torch.from_numpy(b.w.view(np.int32)).cuda()
    g.set_state(w); h.set_state(w); o.set_state(b.w)
    dirty = np.zeros(N, bool)
    for t in range(4, 7):
        if students:
            ro = o.step_students(); rg = g.step_students(); rh = h.step_students()
            so, mo, do = ro[0], ro[1], ro[6]
        else:
            a = o.random_actions(4, t); ta = torch.from_numpy(a).cuda()
            ro = o.step_stacked(a); rg = g.step_stacked(ta); rh = h.step_stacked(ta)
            so, mo, do = ro[0], ro[1], ro[5]
        torch.cuda.synchronize()
        # register kernels vs LDS fallback: the same decisions (masks, cells, done, info, ring stamps and feature counts: exact); the
        # re-projected ranges differ in the last ulp (the two kernels order the rotation arithmetic differently)
        assert float((rg[0] - rh[0]).abs().max()) <= 2e-6, t
        for x, y in zip(rg[1:], rh[1:]):
            assert torch.equal(x, y), t
        wg, wh = g.get_state(), h.get_state()
        rg_ring, rh_ring = o.ring(wg.cpu().numpy().view(np.uint32)), o.ring(wh.cpu().numpy().view(np.uint32))
        live = rg_ring[..., 0] != 0
        np.testing.assert_array_equal(rg_ring[..., 0], rh_ring[..., 0]); np.testing.assert_array_equal(rg_ring[live][:, 1], rh_ring[live][:, 1])
        dirty |= (o.stack_margins() < CELL_MARGIN) | (o.state_margins() < MARGIN)
        sg, mg = rg[0].cpu().numpy(), rg[1].cpu().numpy()
        done = do != 0
        cmp_ = ~dirty & ~done                                 # auto-reset envs show the reset observation on both sides; checked elsewhere
        assert (mg[cmp_] == mo[cmp_]).all(), t
        assert np.abs(sg[cmp_] - so[cmp_]).max() <= OBS_TOL, t
        ro_ring = o.ring()
        ok = ~dirty
        np.testing.assert_array_equal(rg_ring[ok][..., 0], ro_ring[ok][..., 0])
        lv = ro_ring[ok][..., 0] != 0
        np.testing.assert_array_equal(rg_ring[ok][lv][:, 1], ro_ring[ok][lv][:, 1])     # kept features per entry: what the bug lost
    assert (~dirty).sum() > N // 2
    g.close(); h.close(); o.close()


@pytest.mark.parametrize("task", ["level5", "level5_fusion", "level5_dumb"])
def test_persistent_observation_is_bitwise_the_dense_one(task):
    """te_set_persistent_obs: while the caller keeps passing the same stacked buffer, a step only sets the previously patched cells back to
    one and patches the new ones (no 24 KB-per-env background stream).  The buffer must be bit for bit what the dense path writes, through
    auto-resets (reset observation in the main buffer, terminal observation in its own), a buffer swap (dense fallback for that call) and a
    te_observe_stacked in between; N is not a multiple of 64."""
    torch = _gpu()
    from dronechase_amd import default_config
    from dronechase_amd.batched_env import BatchedEnv
    N = 328
    cfg = default_config(task, n_envs=N, motor_noise=1, max_step=9, seed=13)
    students = task == "level5_dumb"
    dense, pers = BatchedEnv(cfg, "cuda:0"), BatchedEnv(cfg, "cuda:0")
    pers.set_persistent_obs(True)
    dense.reset(); pers.reset()

    def both(t):
        if students:
            return dense.step_students(), pers.step_students()
        a = dense.random_actions(6, t)
        return dense.step_stacked(a), pers.step_stacked(a)

    def same(ra, rb, t):
        torch.cuda.synchronize()
        for k, (x, y) in enumerate(zip(ra, rb)):
            assert torch.equal(x, y), (t, k)
        if not students:
            d = dense.done != 0
            if bool(d.any()):
                assert torch.equal(dense.t_stacked[d], pers.t_stacked[d]) and torch.equal(dense.t_mask[d], pers.t_mask[d]), t
    if not students:
        same(dense.observe_stacked(), pers.observe_stacked(), -1)
    dones = 0
    for t in range(36):
        ra, rb = both(t)
        same(ra, rb, t)
        dones += int((dense.done != 0).sum())
        buf = rb[0]
        if t == 5:
            # the mode is really on: a cell nobody patches keeps whatever it holds (a dense step would have streamed a one over it) ...
            flat = buf.view(-1)
            idx = int(torch.nonzero(flat == 1.0)[-1])            # the last empty cell of the buffer
            flat[idx] = 7.0
            ra, rb = both(100 + t)
            torch.cuda.synchronize()
            if float(ra[0].view(-1)[idx]) == 1.0:                # (unless this very step put a feature there)
                assert float(rb[0].view(-1)[idx]) == 7.0, "the persistent path did not run: the background was streamed"
                rb[0].view(-1)[idx] = 1.0
            same(ra, rb, 100 + t)
        if t == 12:   # a different buffer (a rollout slot): garbage in, the dense path must overwrite all of it
            if students: pers._students = (torch.full_like(pers._students[0], 0.5), *pers._students[1:])
            else: pers.stacked = torch.full_like(pers.stacked, 0.5)
        if t == 20 and not students:
            same(dense.observe_stacked(), pers.observe_stacked(), 200)
    assert dones >= N      # every env went through an auto-reset (max_step 9)
    pers.set_persistent_obs(False)
    same(*both(300), 300)
    dense.close(); pers.close()
