#!/usr/bin/env python3
"""The PCIe-inclusive rate: cfg.io_location = TE_IO_HOST, every te_step takes host actions and fills host observation / reward /
done / info buffers (pageable numpy arrays, or pinned ones with --pinned).  Never bench.py's `value`: DESIGN.md §7 quotes it."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dronechase_amd import _lib, config as K, default_config

N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 65536
pinned = "--pinned" in sys.argv
steps, warm = 100, 10
cfg = default_config("stage03", n_envs=N, io_location=K.IO_HOST)
L = _lib.load()
h = C.c_void_p()
_lib.check(L.te_create(C.byref(cfg), 0, C.byref(h)), "te_create")


def buf(shape, dtype):
    t = torch.zeros(shape, dtype=dtype)
    return t.pin_memory() if pinned else t


T = dict(lidar=buf((N, 3, 13, 26), torch.float32), inertial=buf((N, 15), torch.float32), la=buf((N, 4), torch.float32), tl=buf((N, 3, 13, 26), torch.float32),
         ti=buf((N, 15), torch.float32), ta=buf((N, 4), torch.float32), reward=buf((N,), torch.float32), done=buf((N,), torch.uint8),
         info=buf((N, 4), torch.int32), actions=buf((N, 4), torch.float32))
p = {k: C.c_void_p(v.data_ptr()) for k, v in T.items()}
_lib.check(L.te_reset(h, None, None), "te_reset")
for terminal in (True, False):
    t0 = None
    for s in range(warm + steps):
        if s == warm:
            t0 = time.perf_counter()
        _lib.check(L.te_random_actions(h, p["actions"], 1234, s, None), "te_random_actions")
        _lib.check(L.te_step(h, p["actions"], p["lidar"], p["inertial"], p["la"], p["reward"], p["done"], p["info"],
                             p["tl"] if terminal else None, p["ti"] if terminal else None, p["ta"] if terminal else None, None), "te_step")
    dt = (time.perf_counter() - t0) / steps
    mb = N * (4056 + 60 + 16 + 4 + 1 + 16 + 16) / 1e6   # + the terminal rows of the done envs (~1 %)
    print(f"host I/O ({'pinned' if pinned else 'pageable'}), {N} envs, terminal observation {'on' if terminal else 'off'}: {dt * 1e3:.2f} ms/step = "
          f"{N / dt / 1e6:.1f} M env-steps/s ({mb:.0f} MB per step over PCIe = {mb / dt / 1e3:.1f} GB/s)", flush=True)
L.te_destroy(h)
