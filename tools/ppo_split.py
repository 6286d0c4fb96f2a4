#!/usr/bin/env python3
"""BASELINE config 5 (stage03 + PPO) at its own size on ONE MI355X: 65 536 envs, 32-step rollouts (8.7 GB of observations resident in HBM),
minibatches of 65 536 samples.  Prints where the time goes: the environment alone, collect (policy forward + sampling + te_step, HIP graph),
update (10 x 32 minibatches by default: --epochs), for the fp32 / plain-Adam learner and for PPOConfig.fast_learner (fused Adam + bf16 autocast).
    python tools/ppo_split.py [n_envs] [n_steps] [batch] [epochs]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dronechase_amd import default_config
from dronechase_amd.batched_env import BatchedEnv
from dronechase_amd.ppo import PPO, PPOConfig

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32
B = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
E = int(sys.argv[4]) if len(sys.argv) > 4 else 2
out = {"n_envs": N, "n_steps": T, "batch_size": B, "n_epochs": E}


def timed(fn, reps):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


env = BatchedEnv(default_config("stage03", n_envs=N), "cuda:0")
env.reset()
a = env.random_actions(1, 0)
for _ in range(20):
    env.step(a, terminal=False)
t_env = timed(lambda: env.step(a, terminal=False), 200)
out["env_step_us"] = t_env * 1e6
out["env_alone_Msteps_per_s"] = N / t_env / 1e6
env.close()
for fast in (False, True):
    env = BatchedEnv(default_config("stage03", n_envs=N), "cuda:0")
    ppo = PPO(env, PPOConfig(n_steps=T, batch_size=B, n_epochs=E, use_graph=True, fast_learner=fast), seed=3)
    ppo.collect(); ppo.update()          # graph capture, MIOpen algorithm search
    t_col = timed(ppo.collect, 3)
    t_upd = timed(ppo.update, 2)
    stats = ppo.update()
    key = "fast_learner" if fast else "fp32_plain_adam"
    out[key] = {"rollout_buffer_GB": ppo.buf.bytes() / 2 ** 30, "collect_s": t_col, "collect_us_per_step": t_col / T * 1e6,
                "collect_Msteps_per_s": T * N / t_col / 1e6, "update_s": t_upd, "update_ms_per_minibatch": t_upd / (E * ((T * N + B - 1) // B)) * 1e3,
                "update_Msamples_per_s": E * T * N / t_upd / 1e6, "collect_plus_update_Msteps_per_s": T * N / (t_col + t_upd) / 1e6,
                "policy_share_of_collect": 1.0 - t_env * T / t_col, "finite": all(v == v and abs(v) < 1e30 for v in stats.values()), "last_update": stats}
    env.close(); del ppo
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
