#!/usr/bin/env python3
"""Where the sub-step kernel's waves ran and when they finished (needs a stamp build):
    python -c "from dronechase_amd.build import build_library; build_library(force=True, extra_flags=['-DTE_DEBUG_STAMPS', '-DTE_NO_LSTAMP'])"
    python tools/k1_waves.py [N] [steps]
Every workgroup of the last launch left {start, end, HW_ID | XCC_ID << 32, kind}: the flights per SIMD, each SIMD's finish time, and how
much of the launch is the busiest SIMD's queue."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, numpy as np, torch
from dronechase_amd import default_config
from dronechase_amd.batched_env import BatchedEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1100
task = sys.argv[3] if len(sys.argv) > 3 else "stage03"
cfg = default_config(task, n_envs=N)
env = BatchedEnv(cfg, "cuda:0")
env.reset()
a = torch.empty((N, 4), device="cuda")
nchunks = (N + 63) // 64
base = 64 + 16 * (nchunks + 1)
D = cfg.n_drones
n_fill = int(os.environ.get("TE_FILL_WAVES", 512 if cfg.stacked_obs else 256))
n_k1 = n_fill + (D + 2) * nchunks
words = base + 4 * n_k1
buf = (C.c_uint64 * words)()
for i in range(steps):
    env.random_actions(1234, i, out=a)
    if cfg.stacked_obs: env.step_stacked(a)
    else: env.step(a, terminal=True)
    if i in (20, steps // 4, steps - 1):
        torch.cuda.synchronize()
        rc = env.L.te_debug_stamps(env._h, buf, words)
        assert rc == 0
        r = np.frombuffer(buf, dtype=np.uint64)[base:].reshape(n_k1, 4).astype(np.int64)
        kind = r[:, 3]
        fl = (kind == 2) | (kind == 3)
        t0 = r[kind > 0, 0].min()
        start = (r[:, 0] - t0) / 100.0
        end = (r[:, 1] - t0) / 100.0
        hw = r[:, 2] & 0xFFFFFFFF
        xcc = (r[:, 2] >> 32) & 0xF
        simd = (hw >> 4) & 3
        cu = (hw >> 8) & 0xF
        se = (hw >> 12) & 0xF      # SH_ID | SE_ID
        key = ((xcc * 16 + se) * 16 + cu) * 4 + simd
        ukeys, inv = np.unique(key[kind > 0], return_inverse=True)
        nf = np.bincount(inv, weights=fl[kind > 0].astype(float), minlength=len(ukeys)).astype(int)
        fin = np.zeros(len(ukeys))
        np.maximum.at(fin, inv, end[kind > 0])
        print(f"step {i}: launch span {end[kind > 0].max():.1f} us; {int(fl.sum())} flights ({int((kind == 2).sum())} dense, {int((kind == 3).sum())} mixed), "
              f"{int((kind == 1).sum())} fill waves on {len(ukeys)} SIMDs of {len(np.unique(key[kind > 0] // 4))} CUs")
        print(f"    flights per SIMD: mean {nf.mean():.2f}  min {nf.min()}  p50 {np.percentile(nf, 50):.0f}  p90 {np.percentile(nf, 90):.0f}  max {nf.max()}")
        hist = np.bincount(nf)
        print("    histogram (flights: SIMDs) " + "  ".join(f"{k}:{v}" for k, v in enumerate(hist) if v))
        print(f"    flight start p50 {np.percentile(start[fl], 50):.1f} p99 {np.percentile(start[fl], 99):.1f} max {start[fl].max():.1f}; "
              f"duration p50 {np.percentile((end - start)[fl], 50):.1f} p99 {np.percentile((end - start)[fl], 99):.1f} max {(end - start)[fl].max():.1f}")
        print(f"    fill waves: end p50 {np.percentile(end[kind == 1], 50):.1f} max {end[kind == 1].max():.1f}")
        for k in sorted(set(nf.tolist())):
            m = nf == k
            print(f"    SIMDs with {k} flights: {int(m.sum()):4d}, finish mean {fin[m].mean():.1f} max {fin[m].max():.1f} us")
        cu_key = ukeys // 4
        ucu, cinv = np.unique(cu_key, return_inverse=True)
        ncu = np.bincount(cinv, weights=nf.astype(float))
        print(f"    flights per CU: mean {ncu.mean():.1f} min {ncu.min():.0f} max {ncu.max():.0f}")
        xk = ukeys // (4 * 16 * 16)
        ux, xinv = np.unique(xk, return_inverse=True)
        nx = np.bincount(xinv, weights=nf.astype(float))
        print("    flights per XCD: " + " ".join(f"{int(v)}" for v in nx), flush=True)
env.close()
