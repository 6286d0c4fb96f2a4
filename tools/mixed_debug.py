"""Lockstep: mixed waves on (default) vs off (TE_DENSE_MIN=1) must be bit-identical.  Child processes, same seeds."""
import os, subprocess, sys
CHILD = r'''
import sys, torch, numpy as np
sys.path.insert(0, %r)
from dronechase_amd import default_config
from dronechase_amd.batched_env import BatchedEnv
task, N, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
env = BatchedEnv(default_config(task, n_envs=N, seed=5), "cuda:0")
env.reset()
out = []
for t in range(T):
    (env.step_stacked if env.stacked_mode else env.step)(env.random_actions(9, t), terminal=False)
    out.append(env.get_state().cpu().numpy().copy())
np.save(sys.argv[4], np.stack(out))
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
task, N, T = (sys.argv + ["stage03", "256", "300"])[1:4]
for name, extra in (("mixed", {}), ("dense", {"TE_DENSE_MIN": "1"})):
    r = subprocess.run([sys.executable, "-c", CHILD % root, task, N, T, f"/tmp/md_{name}.npy"], env={**os.environ, **extra}, capture_output=True, text=True)
    if r.returncode: print(r.stderr[-2000:]); sys.exit(1)
import numpy as np
sys.path.insert(0, root)
from dronechase_amd import config as K, default_config
a, b = np.load("/tmp/md_mixed.npy"), np.load("/tmp/md_dense.npy")
N = int(N); D = default_config(task, n_envs=1).n_drones
for t in range(a.shape[0]):
    if not np.array_equal(a[t], b[t]):
        da = a[t][: N * D * K.DRONE_WORDS].reshape(N, D, K.DRONE_WORDS); db = b[t][: N * D * K.DRONE_WORDS].reshape(N, D, K.DRONE_WORDS)
        idx = np.argwhere(da != db)
        print("first difference at step", t, "count", len(idx), "first (env, slot, word):", idx[:10].tolist())
        e, s, w = idx[0]
        print("mixed:", da[e, s, [K.D["ARMED"], K.D["NAV_STATE"]]], da[e, s, :8].view(np.float32)); print("dense:", db[e, s, [K.D["ARMED"], K.D["NAV_STATE"]]], db[e, s, :8].view(np.float32))
        armed = da[:, :, K.D["ARMED"]]
        print("armed per slot in that chunk (prev step):", (a[t-1][: N * D * K.DRONE_WORDS].reshape(N, D, K.DRONE_WORDS)[(e // 64) * 64:(e // 64 + 1) * 64, :, K.D["ARMED"]] != 0).sum(0))
        break
else:
    print("bit-identical over", a.shape[0], "steps")
