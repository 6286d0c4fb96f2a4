#!/usr/bin/env python3
"""Throughput harness of the hot path: random-action rollouts of the stage03 environment
(level4 exp03-vFinal: 2 pursuers + 9 invaders, own-sphere LIDAR) on N MI355X.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one env.step() (te_step: two HIP kernels) of every environment of the rank's shard on one batch of
synthetic actions (dir ~ U(-1,1)^3, mag ~ U(0,1), Philox; mirrors apps/threatengage_runner/interactive/analyse.py:55-59)
that te_random_actions left in HBM before the timed region.  Environments shard across ranks with no data-path collective
(weak scaling: --envs-per-gpu is fixed); RNG is keyed on the GLOBAL env index.  Rank 0 prints ONE JSON
line.  Inputs and outputs stay resident in HBM for the whole timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
# VALU issue model of the sub-step kernel ("report VALU utilisation next to HBM %"): static count of
# the loop body, 4 clocks per wave64 VALU instruction, 16 for the quarter-rate ones (tools/loop_cost.py), 1024 SIMDs,
# 2.4 GHz (rocm-smi reads 2.397 GHz for the whole bench)
VALU_CLOCKS_PER_DRONE_SUBSTEP = 1950.0
SIMDS, CLOCK_HZ, SUBSTEPS = 1024, 2.4e9, 16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--task", default="stage03", help="stage01 | stage02 | exp02 | stage03 (= exp03) | exp04 | level5 (stacked observation) | exp05 (ally observed + driven by the caller every step)")
    ap.add_argument("--envs-per-gpu", type=int, default=65536)
    ap.add_argument("--n-invaders", type=int, default=0, help="override I (stage02 with 8 invaders: --n-invaders 8)")
    ap.add_argument("--no-noise", action="store_true", help="motor noise off (parity runs); default on")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--action-seed", type=int, default=1234)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-profile-events", action="store_true", help="do not bracket kernels with HIP events")
    return ap.parse_args()


def host_cores() -> int:
    """Threads the CPU baseline may use: the cgroup CPU quota when there is one, else the affinity mask, capped
    at the 16-core share a one-GPU box gets (oversubscribing 256 visible CPUs would only thrash)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("TE_CPU_THREADS", "16"))))


def cpu_baseline(task: str, overrides: dict, action_seed: int, seconds: float):
    """The oracle (scalar C restatement, float64, OpenMP over envs) timed on this box's host cores on a
    bounded sample of the same workload.  A reported baseline, not the optimisation target."""
    from oracle import te_oracle as O

    cores = host_cores()
    n = 8192
    cfg = O.default_config(task, n_envs=n, **overrides)
    env = O.OracleEnv(cfg, "f64", threads=cores)
    env.reset()
    base_fn = env.step_stacked if cfg.stacked_obs else env.step
    external = cfg.ally_policy == 3  # exp05: the ally's observation is built and its (synthetic) action applied every step

    def fn(a, terminal=False, _n=[0]):
        if external:
            env.observe_ally()
            env.set_ally_actions(env.random_actions(action_seed + 1000, _n[0]))
            _n[0] += 1
        return base_fn(a, terminal=terminal)
    for s in range(3):
        fn(env.random_actions(action_seed, s), terminal=False)
    t0 = time.perf_counter()
    steps = 0
    while True:
        fn(env.random_actions(action_seed, 3 + steps), terminal=False)
        steps += 1
        if (time.perf_counter() - t0 >= seconds and steps >= 10) or steps >= 2000:
            break
    dt = time.perf_counter() - t0
    env.close()
    return {"value": n * steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n} envs x {steps} env-steps of the same task, random actions, motor noise "
                      f"{'on' if cfg.motor_noise else 'off'}, float64 C restatement with OpenMP ({dt:.1f} s); "
                      "PyBullet/PyFlyt are not installable here (no reference build possible)"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    from dronechase_amd import _lib, default_config
    from dronechase_amd.batched_env import BatchedEnv
    from dronechase_amd.build import build_library

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if rank == 0:
        build_library()  # no-op when dronechase_amd/libthreatengage.so is up to date
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: dronechase_amd has no CPU fallback")
    # TE_BENCH_BACKEND=gloo: rehearsal of the multi-rank control flow on a box with fewer GPUs than ranks (ranks share
    # devices round-robin; the reduction runs on host tensors).  The real thing is nccl (= RCCL), one rank per GPU.
    backend = os.environ.get("TE_BENCH_BACKEND", "nccl")
    local_dev = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    device = torch.device("cuda", local_dev)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)  # barrier + max-over-ranks only; no data-path collective
        else:
            dist.init_process_group(backend)
        dist.barrier()

    overrides = dict(motor_noise=0 if args.no_noise else 1, seed=args.seed)
    if args.n_invaders:
        overrides["n_invaders"] = args.n_invaders
    n_local = args.envs_per_gpu
    cfg = default_config(args.task, n_envs=n_local, env_index_base=rank * n_local, **overrides)
    env = BatchedEnv(cfg, device)
    # The inputs of the timed region are resident in HBM when it starts (one [N,4] action batch per step, generated on
    # the device by te_random_actions beforehand): the timed loop is te_step only.  Above 4 GiB of actions the batches
    # are generated step by step inside the loop instead.
    n_total = args.steps + args.warmup
    pregen = n_total * n_local * 16 <= (4 << 30)
    actions = torch.empty((n_total if pregen else 1, n_local, 4), dtype=torch.float32, device=device)
    if pregen:
        for i in range(n_total):
            env.random_actions(args.action_seed, i, out=actions[i])

    step_fn = env.step_stacked if cfg.stacked_obs else env.step   # level5: te_step_stacked (third launch: stacked_kernel)
    # exp05: one env.step = te_observe_ally -> driver -> te_set_ally_actions -> te_step.  The driver here is synthetic
    # (pre-generated random actions: the policy network is the caller's, not this library's), the observation is built
    # every step all the same.
    external = int(cfg.ally_policy) == 3
    ally_actions = None
    if external:
        ally_actions = torch.empty((n_total if pregen else 1, n_local, 4), dtype=torch.float32, device=device)
        if pregen:
            for i in range(n_total):
                env.random_actions(args.action_seed + 1000, i, out=ally_actions[i])

    def one_step(i: int):
        if external:
            env.observe_ally()
            if not pregen:
                env.random_actions(args.action_seed + 1000, i, out=ally_actions[0])
            env.set_ally_actions(ally_actions[i if pregen else 0])
        if pregen:
            step_fn(actions[i], terminal=True)
        else:
            env.random_actions(args.action_seed, i, out=actions[0])
            step_fn(actions[0], terminal=True)

    def armed_per_env() -> float:
        """Mean number of armed drones per env (disarmed slots are not flown: they cost one flag load)."""
        from dronechase_amd import config as K
        w = env.get_state()
        D = cfg.n_drones
        return float((w[: n_local * D * K.DRONE_WORDS].view(n_local, D, K.DRONE_WORDS)[:, :, K.D["ARMED"]] != 0).float().sum(1).mean().item())

    env.reset()
    for i in range(args.warmup):
        one_step(i)
    torch.cuda.synchronize(device)
    use_events = not args.no_profile_events
    n_events = min(args.steps, 64)
    armed_begin = armed_per_env() if rank == 0 else 0.0
    if use_events:  # HIP events bracket the two kernels of the first <= 64 timed steps (each record costs ~3 us of stream time)
        env.profile_begin(n_events)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i)
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    k1_ms, k2_ms, n_prof = env.profile_end() if use_events else (0.0, 0.0, 0)
    done_frac = float(env.done.float().mean().item())
    armed_end = armed_per_env() if rank == 0 else 0.0

    t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        D = cfg.n_drones
        total_env_steps = world * n_local * args.steps
        value = total_env_steps / elapsed
        alg_all = _lib.algorithmic_bytes_per_env_step(cfg)      # whole env.step with all D drones armed = 8108 B for stage03
        # Only ARMED drones are flown (a disarmed slot costs one flag load), and in a random-action rollout most invader
        # slots are empty for the first few hundred steps, so the algorithmic bytes are priced at the armed count that
        # was actually there: the mean of the census taken right before and right after the timed region, weighted to the
        # event-timed window (its first n_events steps).  DESIGN.md 4 has the per-term table.
        frac_window = 0.5 * n_events / max(args.steps, 1)
        armed = armed_begin + (armed_end - armed_begin) * frac_window
        per_drone = 2 * 176                                        # sub-step kernel: state read + written per armed drone
        lidar_bytes = (6 if cfg.stacked_obs else 1) * 3 * 338 * 4     # own sphere, or level5's six stacked spheres
        alg_k1 = armed * per_drone + 16 + lidar_bytes              # + action + LIDAR background
        alg_k2 = alg_all - (D * per_drone + 16 + lidar_bytes)      # engage/observe kernel (+ stacked_kernel in level5): the rest
        alg = alg_k1 + alg_k2
        out = {
            "metric": "env-steps/sec (whole job), random-action rollout",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.task} (level4 exp03-vFinal shape: {cfg.n_pursuers} pursuers + {cfg.n_invaders} "
                                   f"invaders, {'stacked-sphere LIDAR 6x3x13x26' if cfg.stacked_obs else 'own-sphere LIDAR 3x13x26'}), {n_local} envs per GPU, 16 physics sub-steps per env-step, "
                                   f"random actions, motor noise {'on' if cfg.motor_noise else 'off'}, auto-reset on",
                       "task": args.task, "envs_per_gpu": n_local, "total_envs": world * n_local, "drones_per_env": D,
                       "parallelism": f"env-sharded x{world}, no collective"},
            "done_fraction_last_step": done_frac,
        }
        if use_events and n_prof:
            dom_ms, dom_name, dom_bytes = max((k1_ms, "substeps_kernel", alg_k1), (k2_ms, "engage_observe_kernel", alg_k2))
            ach = dom_bytes * n_local / (dom_ms * 1e-3) / 1e9
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
            if os.path.exists(pmc) and args.task in ("stage03", "exp03") and n_local == 65536 and not args.n_invaders:  # what the PMC passes were taken on
                try:
                    traffic = json.load(open(pmc)).get(dom_name, {}).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            out["roofline"] = {"bound": "hbm", "kernel": dom_name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                               "algorithmic_bytes_per_launch": dom_bytes * n_local, "avg_launch_ms": dom_ms,
                               "launches_timed": n_prof, "armed_drones_per_env": armed,
                               "valu_issue_frac_substeps_kernel": (armed * n_local / 64.0) * SUBSTEPS * VALU_CLOCKS_PER_DRONE_SUBSTEP
                                                                  / (SIMDS * CLOCK_HZ * k1_ms * 1e-3),
                               "armed_drones_per_env_begin_end": [armed_begin, armed_end]}
            step_ms = k1_ms + k2_ms
            out["roofline_env_step"] = {"bound": "hbm", "achieved": alg * n_local / (step_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                        "unit": "GB/s", "frac": alg * n_local / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "algorithmic_bytes_per_env_step": alg, "algorithmic_bytes_all_armed": alg_all, "substeps_kernel_ms": k1_ms,
                                        "engage_observe_kernel_ms": k2_ms}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.task, overrides, args.action_seed, args.cpu_seconds)
            except Exception as exc:  # the baseline is a reported extra: never lose the GPU line over it
                out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": 0, "kind": "port", "sample": f"failed: {exc}"}
        print(json.dumps(out), flush=True)
    env.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
