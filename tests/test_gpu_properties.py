"""Size-independent properties of the HIP path at BASELINE.json's full sizes (65 536 stage03 envs,
16 384 stage02 envs with 8 invaders, 4 096 stage01 envs): run on an MI355X with `pytest -m gpu`.

These do not need the oracle (which would take minutes at this size): determinism, shard invariance,
observation invariants, bookkeeping identities between outputs and the state blob."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FULL = [("stage03", 65536, {}), ("stage02", 16384, {"n_invaders": 8}), ("stage01", 4096, {})]


def _torch():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    return torch


def _rollout(env, steps, seed=77, start=0):
    outs = []
    for s in range(steps):
        a = env.random_actions(seed, start + s)
        l, i, la, r, d, info = env.step(a)
        outs.append((l.clone(), i.clone(), r.clone(), d.clone(), info.clone(), env.t_inertial.clone()))
    return outs


@pytest.mark.parametrize("task,N,over", FULL)
def test_determinism_and_shard_invariance(task, N, over):
    """Same seed -> bit-identical outputs; and splitting the env range over two te_env shards (as two
    GPUs would, env_index_base = shard offset) reproduces the single-shard run bit for bit."""
    torch = _torch()
    from dronechase_amd import default_config
    from dronechase_amd.batched_env import BatchedEnv

    T = 12
    full = BatchedEnv(default_config(task, n_envs=N, seed=3, **over), "cuda:0")
    full.reset()
    ref = _rollout(full, T)
    again = BatchedEnv(default_config(task, n_envs=N, seed=3, **over), "cuda:0")
    again.reset()
    for a, b in zip(ref, _rollout(again, T)):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    again.close()
    h = N // 2
    lo = BatchedEnv(default_config(task, n_envs=h, seed=3, env_index_base=0, **over), "cuda:0")
    hi = BatchedEnv(default_config(task, n_envs=N - h, seed=3, env_index_base=h, **over), "cuda:0")
    lo.reset(); hi.reset()
    for s in range(T):
        a_lo, a_hi = lo.random_actions(77, s), hi.random_actions(77, s)
        o_lo, o_hi = lo.step(a_lo), hi.step(a_hi)
        for k, idx in enumerate((0, 1, 3, 4, 5)):  # lidar, inertial, reward, done, info
            got = torch.cat([o_lo[idx], o_hi[idx]])
            assert torch.equal(got, ref[s][(0, 1, 2, 3, 4)[k]]), f"shard mismatch in output {idx} at step {s}"
    # a different seed must change the spawn positions
    other = BatchedEnv(default_config(task, n_envs=256, seed=4, **over), "cuda:0")
    assert not torch.equal(other.reset()[1], full.reset()[1][:256])
    for e in (full, lo, hi, other):
        e.close()


@pytest.mark.parametrize("task,N,over", FULL)
def test_observation_invariants(task, N, over):
    torch = _torch()
    from dronechase_amd import config as K, default_config
    from dronechase_amd.batched_env import BatchedEnv

    cfg = default_config(task, n_envs=N, seed=1, **over)
    env = BatchedEnv(cfg, "cuda:0")
    lidar, inertial, last_action = env.reset()
    assert bool((lidar == 1).all())  # empty sphere after reset
    assert bool((last_action == 0).all())
    D, P = cfg.n_drones, cfg.n_pursuers
    kills_prev = torch.zeros(N, dtype=torch.int32, device="cuda:0")
    any_done = 0
    for s in range(40):
        a = env.random_actions(5, s)
        lidar, inertial, last_action, reward, done, info = env.step(a)
        assert bool(torch.isfinite(reward).all()) and bool(torch.isfinite(inertial).all())
        assert bool(((lidar >= 0) & (lidar <= 1)).all())
        assert bool(((inertial >= -1) & (inertial <= 1)).all())
        hit = lidar[:, 0] < 1
        # flag plane: 0.2 invader / 0.6 wingman on hits, 1 elsewhere; time plane 0.1 on hits
        flag, tim = lidar[:, 1], lidar[:, 2]
        assert bool(((flag[hit] - 0.2).abs().lt(1e-6) | (flag[hit] - 0.6).abs().lt(1e-6)).all())
        assert bool((flag[~hit] == 1).all()) and bool((tim[~hit] == 1).all())
        assert bool((tim[hit] - 0.1).abs().lt(1e-6).all())
        assert bool((hit.flatten(1).sum(1) <= D - 1).all())
        nd = ~done.bool()
        assert bool((last_action[nd] == a[nd]).all())          # last_action echoes the action
        assert bool((last_action[done.bool()] == 0).all())     # reset observation after auto-reset
        assert bool((lidar[done.bool()] == 1).all())
        # hits seen by the agent never exceed the armed drones of the (pre-respawn) state
        st = env.get_state().view(torch.int32)
        dr = st[: N * D * K.DRONE_WORDS].view(N, D, K.DRONE_WORDS)
        armed = dr[:, :, K.D["ARMED"]]
        assert bool(((armed == 0) | (armed == 1)).all())
        assert bool((armed[:, 0] == 1).all())  # the agent is alive at every step boundary (dead => reset)
        if task == "stage03":
            # info[:,0] (agent kills) never decreases within an episode
            same_ep = nd
            assert bool((info[same_ep, 0] >= kills_prev[same_ep]).all())
            kills_prev = torch.where(done.bool(), torch.zeros_like(kills_prev), info[:, 0])
            assert bool(((info[:, 3] >= 1) & (info[:, 3] <= cfg.n_rounds)).all())
            mun = dr[:, :P, K.D["MUNITION"]]
            assert bool(((mun >= 0) & (mun <= cfg.munition)).all())
        any_done += int(done.sum().item())
        # disarmed drones are frozen: zero velocity, zero throttle
        vel = dr[:, :, K.D["VEL"]:K.D["VEL"] + 3].view(torch.float32)
        thr = dr[:, :, K.D["THROTTLE"]:K.D["THROTTLE"] + 4].view(torch.float32)
        dead = armed == 0
        assert bool((vel[dead] == 0).all()) and bool((thr[dead] == 0).all())
        # unit quaternions
        q = dr[:, :, K.D["QUAT"]:K.D["QUAT"] + 4].view(torch.float32)
        assert bool(((q.pow(2).sum(-1) - 1).abs() < 1e-5).all())
    env.close()


@pytest.mark.parametrize("task,N,T", [("stage03", 4096, 400), ("level5", 512, 120), ("evaluation", 2048, 300)])
def test_mixed_waves_fly_exactly_what_dense_waves_would(task, N, T):
    """The sub-step kernel packs the (env, slot) pairs of sparsely armed slots of a chunk into mixed waves (te_env.hip:
    plan_slot / fly<.., MIXED>); TE_DENSE_MIN=1 (read by te_create) gives every armed slot its own wave instead.  Same
    drones, same arithmetic: outputs and state must be bit-identical along a rollout in which sparse slots do occur."""
    import os
    torch = _torch()
    from dronechase_amd import config as K, default_config
    from dronechase_amd.batched_env import BatchedEnv

    cfg = default_config(task, n_envs=N, seed=6)
    mixed = BatchedEnv(cfg, "cuda:0")
    old = os.environ.get("TE_DENSE_MIN")
    os.environ["TE_DENSE_MIN"] = "1"
    try:
        dense = BatchedEnv(cfg, "cuda:0")
    finally:
        if old is None:
            del os.environ["TE_DENSE_MIN"]
        else:
            os.environ["TE_DENSE_MIN"] = old
    mixed.reset(); dense.reset()
    D = cfg.n_drones
    sparse_seen = 0
    for t in range(T):
        a = mixed.random_actions(4, t)
        ra = (mixed.step_stacked if mixed.stacked_mode else mixed.step)(a)
        rb = (dense.step_stacked if dense.stacked_mode else dense.step)(a)
        for x, y in zip(ra, rb):
            assert torch.equal(x, y), (task, t)
        if t % 20 == 0:
            w = mixed.get_state()
            assert torch.equal(w, dense.get_state()), (task, t)
            per_chunk = (w[: N * D * K.DRONE_WORDS].view(N // 64, 64, D, K.DRONE_WORDS)[..., K.D["ARMED"]] != 0).sum(1)
            sparse_seen += int(((per_chunk > 0) & (per_chunk < 40)).sum())
    assert sparse_seen > 0      # the mixed path did carry drones
    mixed.close(); dense.close()


def test_step_writes_the_observation_where_the_caller_wants_it():
    """te_step takes destination pointers: BatchedEnv.step(out=...) / observe(out=...) put the observation straight into
    e.g. slot t of a rollout buffer (dronechase_amd/ppo.py), bit-identical to the internal buffers."""
    torch = _torch()
    from dronechase_amd import default_config
    from dronechase_amd.batched_env import BatchedEnv

    N = 4096
    a_env, b_env = (BatchedEnv(default_config("stage03", n_envs=N, seed=2), "cuda:0") for _ in range(2))
    rollout = (torch.zeros((8, N, 3, 13, 26), device="cuda:0"), torch.zeros((8, N, 15), device="cuda:0"), torch.zeros((8, N, 4), device="cuda:0"))
    a_env.reset(); b_env.reset()
    for x, y in zip(a_env.observe(), b_env.observe(out=tuple(r[0] for r in rollout))):
        assert torch.equal(x, y)
    for t in range(1, 8):
        act = a_env.random_actions(3, 50 + t)
        ra, rb = a_env.step(act), b_env.step(act, out=tuple(r[t] for r in rollout))
        for x, y in zip(ra, rb):
            assert torch.equal(x, y)
        assert rb[0].data_ptr() == rollout[0][t].data_ptr()
    assert bool((rollout[0] <= 1).all()) and bool((rollout[1][1:].abs().sum((1, 2)) > 0).all())   # every slot was written
    with pytest.raises(ValueError, match="16-byte"):
        b_env.step(act, out=(rollout[0][0].view(-1)[1:-1013].view(N - 1, 3, 13, 26)[:0], rollout[1][0], rollout[2][0]))
    a_env.close(); b_env.close()


def test_stage03_episode_statistics():
    """Long random-action rollout at full size: episodes end, waves advance, kills happen, and the
    terminal observation rows are only written for done envs."""
    torch = _torch()
    from dronechase_amd import default_config
    from dronechase_amd.batched_env import BatchedEnv

    N = 65536
    env = BatchedEnv(default_config("stage03", n_envs=N, seed=8), "cuda:0")
    env.reset()
    env.t_inertial.fill_(-7.0)
    dones = 0
    max_wave = 0
    total_kills = 0
    for s in range(400):
        _, _, _, reward, done, info = env.step(env.random_actions(2, s))
        dones += int(done.sum().item())
        max_wave = max(max_wave, int(info[:, 3].max().item()))
    touched = (env.t_inertial != -7.0).any(1)
    assert dones > N // 20 and max_wave >= 3
    assert int(touched.sum().item()) <= dones  # terminal rows only for envs that finished at least once
    assert int(touched.sum().item()) > 0
    env.close()


def test_long_rollout_statistics_match_the_oracle():
    """600 free-running steps of 4 096 stage03 envs on the GPU and in the oracle (same seeds, same actions).  Per-env
    trajectories part ways after the first ambiguous decision (tests/test_gpu_parity.py handles that regime); the
    POPULATION statistics must not: episode ends, kills, waves, rewards within a few percent."""
    import numpy as np
    torch = _torch()
    from dronechase_amd import default_config
    from dronechase_amd.batched_env import BatchedEnv
    from oracle import te_oracle as O
    N, STEPS = 4096, 600
    cfg = default_config("stage03", n_envs=N, motor_noise=1, seed=21)
    g, o = BatchedEnv(cfg, "cuda:0"), O.OracleEnv(cfg, "f32", threads=8)
    g.reset(); o.reset()
    acc = {k: np.zeros(2) for k in ("dones", "reward", "agent_kills", "allies_kills", "deads", "wave")}
    same_done = 0
    for t in range(STEPS):
        a = o.random_actions(77, t)
        *_, ro, do, io = o.step(a)
        *_, rg, dg, ig = g.step(torch.from_numpy(a).cuda())
        dg_, rg_, ig_ = dg.cpu().numpy(), rg.cpu().numpy(), ig.cpu().numpy()
        same_done += int((dg_ == do).sum())
        for k, (x, y) in {"dones": (do.sum(), dg_.sum()), "reward": (ro.sum(), rg_.sum()), "agent_kills": (io[:, 0].sum(), ig_[:, 0].sum()),
                          "allies_kills": (io[:, 1].sum(), ig_[:, 1].sum()), "deads": (io[:, 2].sum(), ig_[:, 2].sum()),
                          "wave": (io[:, 3].sum(), ig_[:, 3].sum())}.items():
            acc[k] += (float(x), float(y))
    assert same_done > 0.97 * N * STEPS, same_done / (N * STEPS)
    assert acc["dones"][0] > 1000         # enough episode ends for the comparison to mean something
    for k, (x, y) in acc.items():
        assert abs(x - y) <= 0.05 * max(abs(x), 1.0), (k, x, y)


@pytest.mark.parametrize("task,over", [("stage03", {}), ("stage02", {"n_invaders": 8}), ("stage01", {}), ("exp03", {"lidar_channels": 2})])
def test_persistent_own_sphere_is_bitwise_the_dense_one(task, over):
    """te_set_persistent_obs with te_step: the sub-step launch's erase waves set the cells of the previous observation back to one and the
    engage kernel patches + records the new ones; no background stream.  Bit for bit the dense path, through auto-resets, a buffer swap
    (dense fallback for that call: step(out=...)), a te_observe in between, and N not a multiple of 64."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    from dronechase_amd import default_config
    from dronechase_amd.batched_env import BatchedEnv
    N = 1000
    cfg = default_config(task, n_envs=N, seed=3, max_step=11, **over)
    dense, pers = BatchedEnv(cfg, "cuda:0"), BatchedEnv(cfg, "cuda:0")
    pers.set_persistent_obs(True)
    dense.reset(); pers.reset()
    other = (torch.full_like(pers.lidar, 0.25), torch.empty_like(pers.inertial), torch.empty_like(pers.last_action))
    dones = hits = 0
    for t in range(40):
        a = dense.random_actions(2, t)
        ra = dense.step(a)
        rb = pers.step(a, out=other) if t in (14, 15, 22) else pers.step(a)     # 14 -> 15: the swapped buffer itself becomes persistent; 22: back and forth
        torch.cuda.synchronize()
        for k, (x, y) in enumerate(zip(ra, rb)):
            assert torch.equal(x, y), (t, k)
        d = dense.done != 0
        if bool(d.any()):
            assert torch.equal(dense.t_lidar[d], pers.t_lidar[d]) and torch.equal(dense.t_inertial[d], pers.t_inertial[d])
        dones += int(d.sum()); hits += int((ra[0][:, 0] < 1).sum())
        if t == 6:   # the mode is on: a cell nobody patches keeps what it holds
            flat = pers.lidar.view(-1)
            idx = int(torch.nonzero(flat == 1.0)[-1])
            flat[idx] = 7.0
            a2 = dense.random_actions(2, 1000)
            ra, rb = dense.step(a2), pers.step(a2)
            torch.cuda.synchronize()
            if float(ra[0].view(-1)[idx]) == 1.0:
                assert float(rb[0].view(-1)[idx]) == 7.0, "the persistent path did not run: the background was streamed"
                rb[0].view(-1)[idx] = 1.0
            assert torch.equal(ra[0], rb[0])
        if t == 25:
            for x, y in zip(dense.observe(), pers.observe()):
                assert torch.equal(x, y)
    assert dones >= N and hits > 1000
    dense.close(); pers.close()
