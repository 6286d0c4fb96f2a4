#!/usr/bin/env python3
"""Exploratory GPU-vs-oracle comparison (prints error statistics; the asserted version is tests/test_gpu_parity.py)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dronechase_amd import config as K, default_config
from dronechase_amd.batched_env import BatchedEnv
from oracle import te_oracle as O

def blob_split(w, N, D):
    dr = w[: N * D * K.DRONE_WORDS].reshape(N, D, K.DRONE_WORDS)
    er = w[N * D * K.DRONE_WORDS:].reshape(N, K.ENV_WORDS)
    return dr, er

def compare(task, N=1024, warm=40, checks=6, noise=0, prec="f32"):
    cfg = default_config(task, n_envs=N, motor_noise=noise, seed=11)
    D = cfg.n_drones
    orc = O.OracleEnv(cfg, prec, threads=8)
    gpu = BatchedEnv(cfg, "cuda:0")
    orc.reset(); gpu.reset()
    # initial state equality
    s0 = orc.get_state(); g0 = gpu.get_state().cpu().numpy().view(np.uint32)
    d0, e0 = blob_split(s0, N, D); dg, eg = blob_split(g0, N, D)
    fw = [i for i in range(K.DRONE_WORDS) if i not in K.D_INT_WORDS]
    print(task, "reset: max |dstate| diff", np.abs(d0[..., fw].view(np.float32) - dg[..., fw].view(np.float32)).max(),
          "int mismatch", (d0[..., list(K.D_INT_WORDS)] != dg[..., list(K.D_INT_WORDS)]).sum(), (e0[:, list(K.E_INT_WORDS)] != eg[:, list(K.E_INT_WORDS)]).sum())
    step = 0
    for chk in range(checks):
        for _ in range(warm):
            orc.step(orc.random_actions(99, step)); step += 1
        st = orc.get_state()
        gpu.set_state(torch.from_numpy(st.view(np.int32)).cuda())
        a = orc.random_actions(99, step); step += 1
        ol, oi, ola, orew, od, oinfo = [x.copy() for x in orc.step(a)]
        otl = orc.t_lidar.copy(); oti = orc.t_inertial.copy()
        marg = orc.margins()
        gl, gi, gla, grew, gd, ginfo = gpu.step(torch.from_numpy(a).cuda())
        torch.cuda.synchronize()
        gl, gi, gla, grew, gd, ginfo = [x.cpu().numpy() for x in (gl, gi, gla, grew, gd, ginfo)]
        gtl = gpu.t_lidar.cpu().numpy(); gti = gpu.t_inertial.cpu().numpy()
        so = orc.get_state(); sg = gpu.get_state().cpu().numpy().view(np.uint32)
        do, eo = blob_split(so, N, D); dg, eg = blob_split(sg, N, D)
        ok = marg > 1e-4
        fdiff = np.abs(do[..., fw].view(np.float32) - dg[..., fw].view(np.float32))
        imis = (do[..., list(K.D_INT_WORDS)] != dg[..., list(K.D_INT_WORDS)]).any(axis=(1, 2)) | (eo[:, list(K.E_INT_WORDS)] != eg[:, list(K.E_INT_WORDS)]).any(axis=1)
        per_env = fdiff.reshape(N, -1).max(1)
        worst_word = np.unravel_index(fdiff[ok & ~imis].argmax(), fdiff[ok & ~imis].shape) if (ok & ~imis).any() else None
        lid = (np.abs(ol - gl).reshape(N, -1).max(1) > 1e-5)
        print(f"  chk{chk} step{step}: done {od.sum()}/{gd.sum()} done-mismatch {(od != gd).sum()} (ambig {(~ok).sum()}) int-mismatch envs {imis.sum()} (outside ambig {(imis & ok).sum()})"
              f" | state max {per_env[ok & ~imis].max():.2e} worst {worst_word} | reward max {np.abs(orew - grew)[ok & ~imis].max():.2e}"
              f" | inertial {np.abs(oi - gi)[ok & ~imis].max():.2e} lidar-mismatch envs {lid.sum()} (outside ambig {(lid & ok & ~imis).sum()})"
              f" | term lidar {np.abs(otl - gtl)[od.astype(bool) & gd.astype(bool)].max() if od.any() else 0:.2e}")
    gpu.close(); orc.close()

if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    for task in (sys.argv[1:] or ["exp03", "exp02", "stage02", "stage01"]):
        for noise in (0, 1):
            print("== noise", noise)
            compare(task, noise=noise)
