#!/usr/bin/env python3
"""Time K2 (engage_observe_kernel) with and without the LIDAR tile stream to separate logic from streaming."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, torch
from dronechase_amd import default_config, _lib
from dronechase_amd.batched_env import BatchedEnv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = BatchedEnv(default_config("stage03", n_envs=N), "cuda:0")
L = env.L
a = torch.empty((N, 4), device="cuda:0")
def run(lidar, terminal, steps=100):
    env.reset()
    for i in range(20):
        env.random_actions(1, i, out=a); env.step(a)
    env.profile_begin(steps)
    p = env._p
    for i in range(steps):
        env.random_actions(1, 20 + i, out=a)
        _lib.check(L.te_step(env._h, p(a), p(env.lidar if lidar else None), p(env.inertial), p(env.last_action), p(env.reward),
                             p(env.done), p(env.info), p(env.t_lidar if terminal else None), p(env.t_inertial if terminal else None),
                             p(env.t_last_action if terminal else None), env._stream()), "te_step")
    return env.profile_end()
for lidar, term in ((True, True), (False, False), (True, False)):
    k1, k2, n = run(lidar, term)
    print(f"lidar={lidar} terminal={term}: K1 {k1*1e3:.1f} us  K2 {k2*1e3:.1f} us")
