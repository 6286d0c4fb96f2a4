"""engage_slots_kernel (te_engage_slots.hpp: one wave per (chunk, drone slot), a workgroup per chunk; the small-shard form of the
engage/observe step) must write bit for bit what engage_kernel<2, 9> (one lane per env, one wave per chunk) writes: every output of every
step, the terminal buffers of the done envs, and the whole state blob, along free-running rollouts with auto-resets, wave advances, shots,
explosions, ragged N, the persistent observation, and the task switches of the level4 family.  TE_ENGAGE=slots / regs selects the kernel
when the te_env is created."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(monkeypatch, task, n, **over):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    from dronechase_amd import default_config
    from dronechase_amd.batched_env import BatchedEnv
    envs = []
    for mode in ("slots", "regs"):
        monkeypatch.setenv("TE_ENGAGE", mode)
        envs.append(BatchedEnv(default_config(task, n_envs=n, **over), "cuda:0"))
    monkeypatch.delenv("TE_ENGAGE")
    return envs


def _same_rollout(a, b, steps, seed=11, persistent=False, external=False):
    import torch
    if persistent:
        a.set_persistent_obs(True); b.set_persistent_obs(True)
    ra, rb = a.reset(), b.reset()
    for x, y in zip(ra, rb):
        assert torch.equal(x, y)
    n_done = 0
    for s in range(steps):
        act = a.random_actions(seed, s)
        if external:
            for e in (a, b):
                e.observe_ally(); e.set_ally_actions(a.random_actions(seed + 1000, s))
        oa, ob = a.step(act, terminal=True), b.step(act, terminal=True)
        for k, (x, y) in enumerate(zip(oa, ob)):
            assert torch.equal(x, y), f"output {k} differs at step {s}"
        d = oa[4].bool()
        n_done += int(d.sum().item())
        if d.any():
            for name in ("t_lidar", "t_inertial", "t_last_action"):
                assert torch.equal(getattr(a, name)[d], getattr(b, name)[d]), f"{name} differs at step {s}"
        if s % 8 == 7 or s == steps - 1:
            assert torch.equal(a.get_state(), b.get_state()), f"state differs after step {s}"
    return n_done


@pytest.mark.parametrize("task,n,over", [
    ("stage03", 8192, {}), ("stage03", 1000, {"seed": 5}), ("exp02", 4096, {}), ("exp04", 2048, {}), ("evaluation", 2048, {}),
    ("stage03", 2048, {"lidar_channels": 2}), ("stage03", 2048, {"motor_noise": 0, "quad_preset": 0}),
    # the widest shape the slot waves serve: 4 pursuers (three behaviour-tree allies, the "first armed ally is skipped" rule) + 12 invader slots
    # = 16 waves per workgroup, against engage_kernel<6, 12>; and the smallest shards
    ("exp03", 2048, {"n_pursuers": 4, "n_invaders": 12, "n_rounds": 12, "max_step": 60}), ("stage03", 1, {"max_step": 30}), ("stage03", 63, {"max_step": 30}),
])
def test_slot_waves_equal_the_one_wave_kernel_along_rollouts(monkeypatch, task, n, over):
    a, b = _pair(monkeypatch, task, n, **over)
    n_done = _same_rollout(a, b, 200)
    assert n_done > 0 or n < 4096 or task == "evaluation"
    a.close(); b.close()


@pytest.mark.parametrize("n,over", [(16384, {"n_invaders": 8}), (1000, {"n_invaders": 8, "seed": 4, "max_step": 40}), (2048, {"n_invaders": 5, "max_step": 11}),
                                     (2048, {"n_invaders": 8, "lidar_channels": 2, "motor_noise": 0})])
def test_stage02_slot_waves_equal_the_one_wave_kernel(monkeypatch, n, over):
    """engage_slots_stage02_kernel against engage_stage02_kernel<2, 8>: shots, the suicide rule, explosions, respawns inside the step, auto-resets."""
    a, b = _pair(monkeypatch, "stage02", n, **over)
    n_done = _same_rollout(a, b, 200)
    assert n_done > 0 or "max_step" not in over
    a.close(); b.close()
    a, b = _pair(monkeypatch, "stage02", n, **over)
    _same_rollout(a, b, 60, persistent=True)
    a.close(); b.close()


def test_slot_waves_with_short_episodes_and_the_persistent_observation(monkeypatch):
    """max_step = 9: every env auto-resets every ten steps (terminal tiles, respawns of every slot, the reset observation), next to wave
    advances; then the same with the own sphere updated in place."""
    for persistent in (False, True):
        a, b = _pair(monkeypatch, "stage03", 3000, seed=2, max_step=9)
        assert _same_rollout(a, b, 60, persistent=persistent) > 10000
        a.close(); b.close()


def test_slot_waves_with_a_caller_driven_ally(monkeypatch):
    a, b = _pair(monkeypatch, "exp05", 2048)
    _same_rollout(a, b, 120, external=True)
    a.close(); b.close()


def test_slot_waves_on_the_all_armed_state(monkeypatch):
    """Every slot armed (round 9), invaders on the born sphere: shots, explosions and double credits within a few steps, eleven live waves per chunk."""
    import torch
    from dronechase_amd import config as K
    a, b = _pair(monkeypatch, "stage03", 4096, seed=8)
    a.reset(); b.reset()
    w = a.get_state().clone()
    N, D = 4096, 11
    dr = w[: N * D * K.DRONE_WORDS].view(N, D, K.DRONE_WORDS)
    er = w[N * D * K.DRONE_WORDS: N * (D * K.DRONE_WORDS + K.ENV_WORDS)].view(N, K.ENV_WORDS)
    g = torch.Generator(device="cuda:0"); g.manual_seed(4)
    dead = dr[:, :, K.D["ARMED"]] == 0
    # close to the pursuers (who start at r = 2): within shooting range of some, within explosion range of a few
    pos3 = (torch.rand((N, D, 3), device="cuda:0", generator=g) - 0.5) * 3.0 + torch.tensor([0.0, 1.0, 1.0], device="cuda:0")
    fl = dr.view(torch.float32)
    for k in range(3):
        fl[:, :, K.D["POS"] + k] = torch.where(dead, pos3[:, :, k], fl[:, :, K.D["POS"] + k])
        fl[:, :, K.D["OBS_POS"] + k] = torch.where(dead, pos3[:, :, k], fl[:, :, K.D["OBS_POS"] + k])
    dr[:, :, K.D["ARMED"]] = 1
    er[:, K.E["ROUND"]] = 9
    er[:, K.E["SNAP_MASK"]] = (1 << D) - 1
    a.set_state(w); b.set_state(w)
    kills = 0
    for s in range(40):
        act = a.random_actions(3, s)
        oa, ob = a.step(act, terminal=True), b.step(act, terminal=True)
        for k, (x, y) in enumerate(zip(oa, ob)):
            assert torch.equal(x, y), f"output {k} differs at step {s}"
        kills += int(oa[5][:, 0].sum().item())
    assert torch.equal(a.get_state(), b.get_state()) and kills > 1000
    a.close(); b.close()


def _same_stacked_rollout(a, b, steps, seed=13, persistent=False):
    import torch
    if persistent:
        a.set_persistent_obs(True); b.set_persistent_obs(True)
    a.reset(); b.reset()
    n_done = 0
    for s in range(steps):
        act = a.random_actions(seed, s)
        oa, ob = a.step_stacked(act), b.step_stacked(act)
        for k, (x, y) in enumerate(zip(oa, ob)):
            assert torch.equal(x, y), f"output {k} differs at step {s}"
        d = oa[5].bool()
        n_done += int(d.sum().item())
        if d.any():
            for name in ("t_stacked", "t_mask", "t_inertial", "t_last_action"):
                assert torch.equal(getattr(a, name)[d], getattr(b, name)[d]), f"{name} differs at step {s}"
        if s % 8 == 7 or s == steps - 1:
            assert torch.equal(a.get_state(), b.get_state()), f"state differs after step {s}"
    return n_done


@pytest.mark.parametrize("task,n,over", [
    ("level5", 4096, {}), ("level5", 1000, {"seed": 3, "max_step": 25}), ("level5", 63, {"max_step": 12}),
    ("level5_c1", 4096, {}), ("level5_c1", 1000, {"seed": 9, "max_step": 25}),
    # odd slot counts: the last wave carries one slot only; more wingmen than half the waves
    ("level5", 2048, {"n_invaders": 11, "n_rounds": 11, "max_step": 40}), ("level5", 2048, {"n_pursuers": 7, "n_invaders": 10, "n_rounds": 10, "max_step": 40}),
])
def test_stacked_slot_waves_equal_the_one_wave_kernel(monkeypatch, task, n, over):
    """engage_slots_multi_kernel<1 / 2, false> (the level5 family: wave w carries the slots w, w + W, ...) against engage_kernel<6, 12>: every output of
    te_step_stacked, the terminal buffers and the state blob (snapshot ring included) along rollouts with shots, wave advances and auto-resets."""
    a, b = _pair(monkeypatch, task, n, **over)
    n_done = _same_stacked_rollout(a, b, 120)
    assert n_done > 0 or "max_step" not in over
    a.close(); b.close()
    a, b = _pair(monkeypatch, task, n, **over)
    _same_stacked_rollout(a, b, 40, persistent=True)
    a.close(); b.close()


@pytest.mark.parametrize("task,spw,over", [("level5", 3, {"max_step": 30}), ("level5_c1", 2, {"max_step": 30}), ("level5_c1", 3, {}),
                                            ("level5", 3, {"n_pursuers": 7, "n_invaders": 13, "n_rounds": 13, "max_step": 40})])
def test_stacked_slot_waves_with_more_slots_per_wave(monkeypatch, task, spw, over):
    """Large shards give a wave more slots (fewer waves per workgroup, every chunk resident at once): TE_SLOT_SPW forces that form on a small one."""
    monkeypatch.setenv("TE_SLOT_SPW", str(spw))
    a, b = _pair(monkeypatch, task, 3000, **over)
    monkeypatch.delenv("TE_SLOT_SPW")
    n_done = _same_stacked_rollout(a, b, 100)
    assert n_done > 0 or "max_step" not in over
    a.close(); b.close()


@pytest.mark.parametrize("task,n,over", [
    ("stage03", 4096, {}), ("stage03", 1000, {"seed": 5, "max_step": 25}), ("exp02", 2048, {}), ("evaluation", 2048, {}), ("stage03", 63, {"max_step": 12}),
    ("stage03", 2048, {"lidar_channels": 2}), ("exp03", 2048, {"n_pursuers": 4, "n_invaders": 12, "n_rounds": 12, "max_step": 60}),
])
def test_two_slots_per_wave_with_the_own_sphere(monkeypatch, task, n, over):
    """engage_slots_multi_kernel<2, true> (what a large level4 shard runs: six waves of two slots instead of eleven of one) against
    engage_kernel, forced onto small shards: outputs, terminal buffers, state; then with the persistent observation."""
    monkeypatch.setenv("TE_SLOT_SPW", "2")
    a, b = _pair(monkeypatch, task, n, **over)
    n_done = _same_rollout(a, b, 160)
    assert n_done > 0 or "max_step" not in over
    a.close(); b.close()
    a, b = _pair(monkeypatch, task, n, **over)
    monkeypatch.delenv("TE_SLOT_SPW")
    _same_rollout(a, b, 60, persistent=True)
    a.close(); b.close()


@pytest.mark.parametrize("n,over", [(4096, {}), (1000, {"seed": 2, "max_step": 40})])
def test_level5_2bt_slot_waves_equal_the_one_wave_kernel(monkeypatch, n, over):
    """level5_2bt (2 behaviour-tree wingmen + 30 invader slots, own sphere, evaluation rules): sixteen waves of two slots against engage_kernel<7, 30>."""
    a, b = _pair(monkeypatch, "level5_2bt", n, **over)
    _same_rollout(a, b, 200)
    a.close(); b.close()
    a, b = _pair(monkeypatch, "level5_2bt", n, **over)
    _same_rollout(a, b, 60, persistent=True)
    a.close(); b.close()


def _pair_var(monkeypatch, var, values, task, n, **over):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    from dronechase_amd import default_config
    from dronechase_amd.batched_env import BatchedEnv
    envs = []
    for v in values:
        monkeypatch.setenv(var, v)
        envs.append(BatchedEnv(default_config(task, n_envs=n, **over), "cuda:0"))
    monkeypatch.delenv(var)
    return envs


@pytest.mark.parametrize("task,n,over", [("level5", 2048, {"max_step": 30}), ("level5", 1000, {"seed": 7}), ("level5_c1", 2048, {"max_step": 30}),
                                          ("level5", 63, {"max_step": 12})])
def test_ring_push_dealt_over_four_waves_equals_the_one_wave_push(monkeypatch, task, n, over):
    """ring_push_kernel<18, 4> (small shards: a (chunk, wingman) pair's binning dealt over four waves, LDS hand-over, wave 0 writes the entry)
    against ring_push_kernel<18, 1>: stacked observation, masks, terminal buffers and the state blob with the snapshot ring."""
    a, b = _pair_var(monkeypatch, "TE_PUSH_SPLIT", ("1", "0"), task, n, **over)
    n_done = _same_stacked_rollout(a, b, 100)
    assert n_done > 0 or "max_step" not in over
    a.close(); b.close()


@pytest.mark.parametrize("task,n,over", [("level5_fusion", 4096, {}), ("level5_fusion", 1000, {"seed": 4, "max_step": 30}), ("level5_dumb", 4096, {}),
                                          ("level5_dumb", 1000, {"seed": 6, "max_step": 30}), ("level5_dumb", 63, {"max_step": 12})])
def test_wide_slot_waves_equal_the_one_wave_kernel(monkeypatch, task, n, over):
    """engage_slots_multi_kernel<3, false, WIDE> (36 / 37 drones: thirteen waves of three slots, 64-bit slot masks) against engage_kernel<7, 30>."""
    a, b = _pair(monkeypatch, task, n, **over)
    n_done = _same_stacked_rollout(a, b, 120)
    assert n_done > 0 or "max_step" not in over
    a.close(); b.close()
    a, b = _pair(monkeypatch, task, n, **over)
    _same_stacked_rollout(a, b, 40, persistent=True)
    a.close(); b.close()
