"""cfg.io_location = TE_IO_HOST: the C ABI with HOST pointers — numpy arrays straight into te_reset / te_step /
te_observe / te_random_actions / te_get_state / te_set_state — must produce bit for bit what the device-pointer path produces."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("task,over", [("stage03", {}), ("stage01", {}), ("stage03", {"lidar_channels": 2})])
def test_host_pointers_equal_device_pointers(task, over):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    from dronechase_amd import _lib, config as K, default_config
    from dronechase_amd.batched_env import BatchedEnv
    N = 1000   # not a multiple of 64 on purpose
    dev = BatchedEnv(default_config(task, n_envs=N, seed=5, **over), "cuda:0")
    cfg = default_config(task, n_envs=N, seed=5, io_location=K.IO_HOST, **over)
    L = _lib.load()
    h = C.c_void_p()
    _lib.check(L.te_create(C.byref(cfg), 0, C.byref(h)), "te_create")
    ch = int(cfg.lidar_channels)
    lidar, inertial, la = np.zeros((N, ch, 13, 26), np.float32), np.zeros((N, 15), np.float32), np.zeros((N, 4), np.float32)
    tl, ti, ta = np.zeros_like(lidar), np.zeros_like(inertial), np.zeros_like(la)
    reward, done, info = np.zeros(N, np.float32), np.zeros(N, np.uint8), np.zeros((N, 4), np.int32)
    actions = np.zeros((N, 4), np.float32)
    _lib.check(L.te_reset(h, None, None), "te_reset")
    dev.reset()
    _lib.check(L.te_observe(h, _p(lidar), _p(inertial), _p(la), None), "te_observe")
    dl, di, da = (x.cpu().numpy() for x in dev.observe())
    assert np.array_equal(lidar, dl) and np.array_equal(inertial, di) and np.array_equal(la, da)
    n_done = 0
    for s in range(60):
        _lib.check(L.te_random_actions(h, _p(actions), 77, s, None), "te_random_actions")
        a_dev = dev.random_actions(77, s)
        assert np.array_equal(actions, a_dev.cpu().numpy())
        _lib.check(L.te_step(h, _p(actions), _p(lidar), _p(inertial), _p(la), _p(reward), _p(done), _p(info), _p(tl), _p(ti), _p(ta), None), "te_step")
        dl, di, da, dr, dd, dinf = (x.cpu().numpy() for x in dev.step(a_dev))
        assert np.array_equal(lidar, dl) and np.array_equal(inertial, di) and np.array_equal(la, da), s
        assert np.array_equal(reward, dr) and np.array_equal(done, dd) and np.array_equal(info, dinf), s
        d = done.astype(bool)
        if d.any():
            n_done += int(d.sum())
            assert np.array_equal(tl[d], dev.t_lidar.cpu().numpy()[d]) and np.array_equal(ti[d], dev.t_inertial.cpu().numpy()[d])
    words = C.c_size_t()
    _lib.check(L.te_state_words(h, C.byref(words)), "te_state_words")
    blob = np.zeros(words.value, np.uint32)
    _lib.check(L.te_get_state(h, _p(blob), words.value, None), "te_get_state")
    assert np.array_equal(blob, dev.get_state().cpu().numpy().view(np.uint32))
    # a masked reset through a host mask, then the state goes back in through a host blob
    mask = (np.arange(N) % 3 == 0).astype(np.uint8)
    _lib.check(L.te_reset(h, _p(mask), None), "te_reset"); dev.reset(torch.from_numpy(mask))
    _lib.check(L.te_get_state(h, _p(blob), words.value, None), "te_get_state")
    assert np.array_equal(blob, dev.get_state().cpu().numpy().view(np.uint32))
    _lib.check(L.te_set_state(h, _p(blob), words.value, None), "te_set_state")
    blob2 = np.zeros_like(blob)
    _lib.check(L.te_get_state(h, _p(blob2), words.value, None), "te_get_state")
    assert np.array_equal(blob, blob2)
    assert task == "stage01" or n_done >= 0
    L.te_destroy(h); dev.close()


def test_host_io_rejects_what_it_does_not_serve():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    from dronechase_amd import _lib, config as K, default_config
    L = _lib.load()
    h = C.c_void_p()
    assert L.te_create(C.byref(default_config("level5", n_envs=64, io_location=K.IO_HOST)), 0, C.byref(h)) != 0
    assert b"TE_IO_HOST" in L.te_last_error()
    assert L.te_create(C.byref(default_config("exp03", n_envs=64, lidar_channels=4)), 0, C.byref(h)) != 0


@pytest.mark.parametrize("task", ["stage03", "stage01"])
def test_persistent_observation_under_host_io_is_bitwise_the_dense_one(task):
    """te_set_persistent_obs with HOST pointers: the observation staging of the library is patched in place from step to step, so the
    compaction of the done envs' terminal rows must not land in it (round-3 review: rows 0..n_done-1 kept stale terminal features).
    Two te_envs on host pointers, one persistent, one dense, terminal buffers passed, auto-resets happening: every output equal bit for bit."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    from dronechase_amd import _lib, config as K, default_config
    N = 1500
    L = _lib.load()
    hs = []
    for persistent in (1, 0):
        cfg = default_config(task, n_envs=N, seed=9, io_location=K.IO_HOST)
        cfg.max_step = 12   # episodes end (and auto-reset) every dozen steps: many done rows per step
        h = C.c_void_p()
        _lib.check(L.te_create(C.byref(cfg), 0, C.byref(h)), "te_create")
        _lib.check(L.te_set_persistent_obs(h, persistent), "te_set_persistent_obs")
        _lib.check(L.te_reset(h, None, None), "te_reset")
        hs.append(h)

    def bufs():
        return dict(lidar=np.zeros((N, 3, 13, 26), np.float32), inertial=np.zeros((N, 15), np.float32), la=np.zeros((N, 4), np.float32),
                    tl=np.zeros((N, 3, 13, 26), np.float32), ti=np.zeros((N, 15), np.float32), ta=np.zeros((N, 4), np.float32),
                    reward=np.zeros(N, np.float32), done=np.zeros(N, np.uint8), info=np.zeros((N, 4), np.int32))
    a, b = bufs(), bufs()
    actions = np.zeros((N, 4), np.float32)
    n_done = 0
    for s in range(80):
        _lib.check(L.te_random_actions(hs[0], _p(actions), 31, s, None), "te_random_actions")
        for h, o in ((hs[0], a), (hs[1], b)):
            _lib.check(L.te_step(h, _p(actions), _p(o["lidar"]), _p(o["inertial"]), _p(o["la"]), _p(o["reward"]), _p(o["done"]), _p(o["info"]),
                                 _p(o["tl"]), _p(o["ti"]), _p(o["ta"]), None), "te_step")
        for k in ("lidar", "inertial", "la", "reward", "done", "info"):
            assert np.array_equal(a[k], b[k]), (s, k)
        d = a["done"].astype(bool)
        n_done += int(d.sum())
        for k in ("tl", "ti", "ta"):
            assert np.array_equal(a[k][d], b[k][d]), (s, k)
    assert n_done > 2000
    for h in hs:
        L.te_destroy(h)
