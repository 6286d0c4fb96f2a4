#!/usr/bin/env python3
"""Throughput harness of the hot path: random-action rollouts of the stage03 environment
(level4 exp03-vFinal: 2 pursuers + 9 invaders, own-sphere LIDAR) on N MI355X.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one env.step() (te_step: two HIP kernels) of every environment of the rank's shard on one batch of
synthetic actions (dir ~ U(-1,1)^3, mag ~ U(0,1), Philox; mirrors apps/threatengage_runner/interactive/analyse.py:55-59)
that te_random_actions left in HBM before the timed region.  Environments shard across ranks with no data-path collective;
RNG is keyed on the GLOBAL env index.  The headline is the configuration BASELINE.json's metric names: 65 536 stage03 envs IN
TOTAL (--total-envs), i.e. 65 536 / N per GPU — strong scaling; with N > 1 the line also carries a `weak_scaling` block
(65 536 envs PER GPU, the size one MI355X is efficient at).  --envs-per-gpu fixes the shard size instead (then `scaling` says
"weak").  Rank 0 prints ONE JSON line.  Inputs and outputs stay resident in HBM for the whole timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
# VALU utilisation of the sub-step kernel ("report VALU utilisation next to HBM %") comes from the committed counters
# (profiles/pmc_valu_latest.json, tools/pmc_valu.sh: SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs against GRBM_GUI_ACTIVE / 8), like `traffic` from
# profiles/pmc_latest.json.  (Rounds 1-3 printed a static model, 1 950 clocks per armed-drone sub-step = 0.81 of the launch; the counters read
# 1 351 busy cycles per armed-drone sub-step = 0.51: the model priced the quarter-rate instructions at 16 clocks, the hardware issues them in 8.)
SIMDS, CLOCK_HZ, SUBSTEPS = 1024, 2.4e9, 16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--task", default="stage03", help="stage01 | stage02 | exp02 | stage03 (= exp03) | exp04 | level5 (stacked observation) | exp05 (ally observed + driven by the caller every step) | evaluation | level5_2bt")
    ap.add_argument("--total-envs", type=int, default=65536, help="envs of the whole job, split evenly over the ranks (BASELINE.json: 65 536)")
    ap.add_argument("--envs-per-gpu", type=int, default=0, help="fix the shard size instead of the total (weak scaling)")
    ap.add_argument("--no-weak-block", action="store_true", help="N > 1: skip the extra 65 536-envs-per-GPU measurement")
    ap.add_argument("--n-invaders", type=int, default=0, help="override I (stage02 with 8 invaders: --n-invaders 8)")
    ap.add_argument("--no-noise", action="store_true", help="motor noise off (parity runs); default on")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--action-seed", type=int, default=1234)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-profile-events", action="store_true", help="do not time the kernels with HIP events (a separate short loop after each window)")
    ap.add_argument("--headline-only", action="store_true", help="skip the steady-state and all-armed windows")
    ap.add_argument("--steady-after", type=int, default=1000, help="rollout step at which the steady-state window starts")
    ap.add_argument("--steady-steps", type=int, default=200)
    ap.add_argument("--persistent-obs", action="store_true", help="level5 family: te_set_persistent_obs (the stacked observation is updated in place: the caller passes the same buffer every step)")
    ap.add_argument("--own-stream", action="store_true", help="drive te_step from a stream of the bench's own instead of torch's current (null) stream")
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


def free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args) -> int:
    """`python3 bench.py --gpus N` launched plainly (no WORLD_SIZE in the environment) with N > 1: start the N ranks as a FRESH child
    (`python -m torch.distributed.run`, one rank per GPU) before this process has touched the GPU, let its rank 0 print the one JSON line
    on our stdout, and return its exit code.  Never an exec: a child process."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this pool
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    # read by the ROCm runtime / RCCL when they initialise: set before torch touches the GPU
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this pool
    os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "1")     # an RCCL collective that times out raises here (and the run falls back to gloo) instead of aborting the rank
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
    # Control plane = barrier + max-over-ranks time; there is NO data-path collective (envs shard embarrassingly).  The default
    # process group is gloo (host tensors: it cannot lose the run); on top of it one rank per GPU opens an RCCL ("nccl") group,
    # proves it with one all-reduce, and from then on the barriers of the timed windows run over RCCL / xGMI.  If RCCL does not
    # come up on every rank the run continues on gloo and the line says so (`control_plane`).
    # TE_BENCH_BACKEND=gloo: rehearsal of the multi-rank control flow on a box with fewer GPUs than ranks (ranks share devices
    # round-robin, no RCCL group).
    backend = os.environ.get("TE_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    local_dev = local_rank % n_dev
    torch.cuda.set_device(local_dev)
    device = torch.device("cuda", local_dev)
    rccl, rccl_broken, control_plane = None, False, "single process"
    if world > 1:
        import datetime
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=600))
        control_plane = "gloo (TE_BENCH_BACKEND=gloo)"
        if backend == "nccl" and world > n_dev:
            control_plane = f"gloo ({world} ranks share {n_dev} GPU(s): a rehearsal, RCCL needs one GPU per rank)"
        elif backend == "nccl":
            ok, why = 1, ""
            try:
                rccl = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120))
                probe = torch.ones(1, device=device)
                dist.all_reduce(probe, group=rccl)
                torch.cuda.synchronize(device)
                ok = int(probe.item() == world)
            except Exception as exc:  # noqa: BLE001 - any RCCL failure falls back to the gloo control plane
                ok, why = 0, f"{type(exc).__name__}: {exc}"[:200]
            agree = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(agree, op=dist.ReduceOp.MIN)
            if int(agree.item()) == 1:
                control_plane = "rccl (nccl backend: barrier + max-over-ranks; gloo default group underneath)"
            else:
                rccl, rccl_broken = None, True
                control_plane = f"gloo (RCCL group did not come up on every rank{': ' + why if why else ''})"
        dist.barrier()

    def barrier():
        if world > 1:
            if rccl is not None:
                dist.barrier(group=rccl, device_ids=[local_dev])
            else:
                dist.barrier()

    def max_over_ranks(x: float) -> float:
        if world == 1:
            return x
        if rccl is not None:
            t = torch.tensor([x], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=rccl)
        else:
            t = torch.tensor([x], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    overrides = dict(motor_noise=0 if args.no_noise else 1, seed=args.seed)
    if args.n_invaders:
        overrides["n_invaders"] = args.n_invaders
    use_events = not args.no_profile_events
    markers = os.environ.get("TE_PROF") == "markers"

    def measure(n_local: int, headline_only: bool, scaling: str) -> dict:
        """Everything measured on one shard size: the headline window, and (unless headline_only) the steady-state and all-armed regimes.
        Returns the JSON line's fields (meaningful on rank 0)."""
        cfg = default_config(args.task, n_envs=n_local, env_index_base=rank * n_local, **overrides)
        env = BatchedEnv(cfg, device)
        if args.persistent_obs:
            env.set_persistent_obs(True)
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

        n_batches = n_total if pregen else 1

        def timed(first: int, count: int) -> float:
            """Wall time of `count` consecutive steps starting at rollout step `first`: barrier + synchronize, the steps, synchronize — the
            rank's clock stops THERE; the closing barrier and the max-over-ranks reduction come after it (a control-plane barrier costs tens
            of microseconds, a 20-step window at a small shard is under a millisecond).  Nothing but the step's own launches is enqueued
            in between (no event records)."""
            barrier()
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for i in range(count):
                one_step((first + i) % n_batches if pregen else first + i)
            torch.cuda.synchronize(device)
            el = time.perf_counter() - t0
            barrier()
            return max_over_ranks(el)

        def kernel_times(first: int, count: int):
            """Average duration of the two kernels over the `count` steps starting at rollout step `first`: HIP events on the stream te_step
            launches on (te_profile_begin/end).  The events are the kernels' own start / stop events (hipExtLaunchKernel: the dispatch's begin /
            end timestamps, the quantity rocprofv3 --kernel-trace reports), so nothing is subtracted.  A SEPARATE loop (regime() replays the
            window it timed on the wall clock from a saved state): the headline loop must not pay for event records."""
            env.profile_begin(count)
            for i in range(count):
                one_step((first + i) % n_batches if pregen else first + i)
            torch.cuda.synchronize(device)
            return env.profile_end()

        D = cfg.n_drones
        per_drone = 2 * 176                                           # sub-step kernel: state read + written per armed drone
        lidar_bytes = (6 if cfg.stacked_obs else 1) * 3 * 338 * 4     # own sphere, or level5's six stacked spheres
        alg_all = _lib.algorithmic_bytes_per_env_step(cfg)            # every slot armed (8108 B for stage03)
        alg_k2 = alg_all - (D * per_drone + 16 + lidar_bytes)         # engage/observe kernel (+ stacked_kernel in level5): the rest

        def priced(armed: float):
            """Algorithmic bytes of one env-step with `armed` drones flown: only ARMED drones are flown (a disarmed slot costs one
            scalar flag load), so the drone-state term of SURVEY.md 8(d) is priced at the census, not at all D slots."""
            k1 = armed * per_drone + 16 + lidar_bytes
            return k1, k1 + alg_k2

        timed_in = ("a replay of the same steps from the state saved at the start of the window; HIP events on te_step's stream = "
                    + ("marker packets between the launches (TE_PROF=markers: each bracket includes ~5 us of queue time)" if markers else
                       "the kernels' own start / stop events (hipExtLaunchKernel: dispatch begin / end timestamps, as in rocprofv3 --kernel-trace)"))

        def regime(first: int, count: int, n_events: int, label: str):
            """One measured regime of the rollout: `count` steps on the wall clock (no events), then — unless n_events is 0 — the same
            `count` steps once more from the saved state with the kernels timed by HIP events.  Every fraction of the HBM peak is
            computed from the WALL time of the step."""
            a0 = armed_per_env() if rank == 0 else 0.0
            replay = use_events and n_events > 0
            saved = env.get_state().clone() if replay else None
            el = timed(first, count)
            a1 = armed_per_env() if rank == 0 else 0.0
            k1_ms = k2_ms = 0.0; n_prof = 0
            if replay:   # the SAME steps again from the saved state (the rollout is deterministic), this time with the kernels timed
                env.set_state(saved)
                k1_ms, k2_ms, n_prof = kernel_times(first, count)
                del saved
            armed = 0.5 * (a0 + a1)
            ms = 1e3 * el / count
            b_k1, b_step = priced(armed)
            out = {"label": label, "first_step": first, "steps": count, "value": world * n_local * count / el, "unit": "env-steps/s",
                   "ms_per_step": ms, "armed_drones_per_env": armed, "armed_drones_per_env_begin_end": [a0, a1],
                   "algorithmic_bytes_per_env_step": b_step,
                   "roofline_env_step": {"bound": "hbm", "achieved": b_step * n_local / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": b_step * n_local / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "from": "wall-clock ms_per_step"}}
            if n_prof:
                out["kernels"] = {"substeps_kernel_ms": k1_ms, "engage_observe_kernel_ms": k2_ms, "launches_timed": n_prof, "timed_in": timed_in,
                                  "substeps_kernel_hbm_frac": b_k1 * n_local / (k1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "engage_observe_kernel_hbm_frac": alg_k2 * n_local / (k2_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            return out, el, (k1_ms, k2_ms, n_prof, armed)

        env.reset()
        for i in range(args.warmup):
            one_step(i)
        torch.cuda.synchronize(device)
        # ---- headline: EXACTLY args.steps steps after args.warmup, nothing else in the timed region
        head, elapsed, (k1_ms, k2_ms, n_prof, armed) = regime(args.warmup, args.steps, args.steps, "headline")
        done_frac = float(env.done.float().mean().item())
        extra = {}
        if not headline_only:
            # ---- steady state: the rollout gets heavier as episodes progress (more invaders per wave, spread over more slots);
            # fast-forward to step args.steady_after (untimed), then time args.steady_steps steps the same way
            pos = args.warmup + args.steps   # (the event replay of a window ends where the window ended)
            for i in range(pos, max(pos, args.steady_after)):
                one_step(i % n_batches if pregen else i)
            pos = max(pos, args.steady_after)
            extra["steady_state"], _, _ = regime(pos, args.steady_steps, 32, f"steady state: steps {pos}..{pos + args.steady_steps} of the same rollout")
            # ---- every slot armed: the heaviest step the task can ask for (a trained policy in the last waves).  A state blob
            # with all D drones armed is loaded (te_set_state) and a short window is timed before episodes end and thin it out
            if int(cfg.task) in (3, 4, 5, 7) and not cfg.stacked_obs and not cfg.evaluation:
                from dronechase_amd import config as K
                w = env.get_state().clone()
                dr = w[: n_local * D * K.DRONE_WORDS].view(n_local, D, K.DRONE_WORDS)
                er = w[n_local * D * K.DRONE_WORDS: n_local * (D * K.DRONE_WORDS + K.ENV_WORDS)].view(n_local, K.ENV_WORDS)
                g = torch.Generator(device=device); g.manual_seed(99)
                # disarmed invaders are put on the born sphere like a new wave would (exp03_vFinal_task.py:584-608), at rest
                dead = dr[:, :, K.D["ARMED"]] == 0
                th = torch.rand((n_local, D), device=device, generator=g) * 3.14159265
                ph = 0.8411 + torch.rand((n_local, D), device=device, generator=g) * (1.5707963 - 0.8411)
                pos3 = torch.stack((6 * ph.sin() * th.cos(), 6 * ph.sin() * th.sin(), 6 * ph.cos()), -1)
                fl = dr.view(torch.float32)
                for k in range(3):
                    fl[:, :, K.D["POS"] + k] = torch.where(dead, pos3[:, :, k], fl[:, :, K.D["POS"] + k])
                    fl[:, :, K.D["OBS_POS"] + k] = torch.where(dead, pos3[:, :, k], fl[:, :, K.D["OBS_POS"] + k])
                dr[:, :, K.D["ARMED"]] = 1
                er[:, K.E["ROUND"]] = int(cfg.n_rounds)
                er[:, K.E["SNAP_MASK"]] = (1 << D) - 1
                env.set_state(w)
                extra["all_armed"], _, _ = regime(0, 48, 16, f"every slot armed (round {int(cfg.n_rounds)} state loaded with te_set_state), first 48 steps")

        out = {}
        if rank == 0:
            value = world * n_local * args.steps / elapsed
            b_k1, alg = priced(armed)
            out = {
                "metric": "env-steps/sec (whole job), random-action rollout",
                "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": scaling,
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"{args.task} (level4 exp03-vFinal shape: {cfg.n_pursuers} pursuers + {cfg.n_invaders} "
                                       f"invader slots, of which {armed:.1f} drones per env are armed (flown) in the timed window; "
                                       f"{'stacked-sphere LIDAR 6x3x13x26' if cfg.stacked_obs else 'own-sphere LIDAR 3x13x26'}), {world * n_local} envs in total = {n_local} envs per GPU, "
                                       f"16 physics sub-steps per env-step, random actions, motor noise {'on' if cfg.motor_noise else 'off'}, auto-reset on",
                           "task": args.task, "envs_per_gpu": n_local, "total_envs": world * n_local, "drone_slots_per_env": D,
                           "armed_drones_per_env": armed, "parallelism": f"env-sharded x{world}, no collective", "control_plane": control_plane,
                           "persistent_obs": bool(args.persistent_obs)},
                "done_fraction_last_step": done_frac,
            }
            if n_prof:
                dom_ms, dom_name, dom_bytes = max((k1_ms, "substeps_kernel", b_k1), (k2_ms, "engage_observe_kernel", alg_k2))
                ach = dom_bytes * n_local / (dom_ms * 1e-3) / 1e9
                traffic, traffic_source = None, None
                pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
                if os.path.exists(pmc) and args.task in ("stage03", "exp03") and n_local == 65536 and not args.n_invaders:  # what the PMC passes were taken on
                    try:
                        rec = json.load(open(pmc))
                        traffic = rec.get(dom_name, {}).get("hbm_bytes_per_launch")
                        traffic_source = f"profiles/pmc_latest.json ({rec.get('source', 'separate rocprofv3 --pmc passes of this command')}): not measured in this run"
                    except Exception:
                        traffic = None
                valu = {}
                try:   # the VALU side of the same launch, from the committed counters (never a model)
                    rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_valu_latest.json")))
                    k1 = rec["substeps_kernel"]
                    cyc = k1["valu_busy_cycles_per_drone_substep"]
                    valu = {"valu_busy_frac_substeps_kernel_pmc": k1["valu_busy_frac"],   # measured: the profiled command (stage03, 65 536 envs, driver window)
                            "valu_busy_cycles_per_drone_substep_pmc": cyc, "valu_instructions_per_drone_substep_pmc": k1["valu_instructions_per_drone_substep"],
                            # this run's launch priced with the measured cycles per armed-drone sub-step, at the 2.4 GHz the chip holds in this bench
                            "valu_busy_frac_substeps_kernel": (armed * n_local / 64.0) * SUBSTEPS * cyc / (SIMDS * CLOCK_HZ * k1_ms * 1e-3),
                            "valu_source": "profiles/pmc_valu_latest.json (" + rec.get("source", "") + "): not measured in this run"}
                except Exception:
                    pass
                out["roofline"] = {"bound": "hbm", "kernel": dom_name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                                   "algorithmic_bytes_per_launch": dom_bytes * n_local, "avg_launch_ms": dom_ms,
                                   "launches_timed": n_prof, "timed_in": timed_in,
                                   "armed_drones_per_env": armed, **valu,
                                   "armed_drones_per_env_begin_end": head["armed_drones_per_env_begin_end"]}
                # the other kernel of the step next to it (round-3 review: its fraction and traffic ratio belong in the headline too)
                oth_ms, oth_name, oth_bytes = min((k1_ms, "substeps_kernel", b_k1), (k2_ms, "engage_observe_kernel", alg_k2))
                oth = {"kernel": oth_name, "avg_launch_ms": oth_ms, "algorithmic_bytes_per_launch": oth_bytes * n_local,
                       "frac": oth_bytes * n_local / (oth_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
                if traffic is not None:
                    try:
                        t2 = json.load(open(pmc)).get(oth_name, {}).get("hbm_bytes_per_launch")
                        if t2:
                            oth["traffic"] = t2; oth["traffic_over_algorithmic"] = t2 / (oth_bytes * n_local)
                    except Exception:
                        pass
                out["roofline"]["other_kernel"] = oth
            step_ms = 1e3 * elapsed / args.steps
            out["roofline_env_step"] = {"bound": "hbm", "achieved": alg * n_local / (step_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                        "unit": "GB/s", "frac": alg * n_local / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "from": "wall-clock ms_per_step of the headline window (per GPU)",
                                        "algorithmic_bytes_per_env_step": alg, "algorithmic_bytes_all_armed": alg_all, "substeps_kernel_ms": k1_ms,
                                        "engage_observe_kernel_ms": k2_ms}
            out.update(extra)
        env.close()
        del env, actions
        torch.cuda.empty_cache()
        return out

    # te_step enqueues on the caller's stream (BatchedEnv passes torch's current one): the legacy null stream unless --own-stream.
    # Interleaved A/B on one box (round 3): 851 / 851 / 861 M env-steps/s on the null stream against 848 / 855 / 846 M on a
    # stream of the bench's own in the driver's window — no difference.
    import contextlib
    lane = torch.cuda.stream(torch.cuda.Stream(device)) if args.own_stream else contextlib.nullcontext()
    with lane:
        if args.envs_per_gpu:       # an explicit shard size: per-GPU work fixed as N grows
            out = measure(args.envs_per_gpu, args.headline_only, "weak")
        else:                       # BASELINE.json's configuration: --total-envs in total, split over the ranks
            if args.total_envs % world:
                raise SystemExit(f"--total-envs {args.total_envs} does not split over {world} ranks")
            out = measure(args.total_envs // world, args.headline_only, "strong")   # the metric's configuration: total fixed, at every N incl. 1
            if world > 1 and not args.no_weak_block:
                # the size one MI355X is efficient at, on every rank: what the node delivers when the caller has 65 536 envs PER GPU
                wk = measure(args.total_envs, True, "weak")
                if rank == 0:
                    out["weak_scaling"] = {k: wk[k] for k in ("value", "unit", "ms_per_step", "scaling", "config", "roofline", "roofline_env_step") if k in wk}
        torch.cuda.current_stream(device).synchronize()
    if rank == 0:
        out.setdefault("config", {})["stream"] = "a non-default HIP stream (torch.cuda.Stream)" if args.own_stream else "torch's current stream (the legacy null stream)"
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.task, overrides, args.action_seed, args.cpu_seconds)
            except Exception as exc:  # the baseline is a reported extra: never lose the GPU line over it
                out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": 0, "kind": "port", "sample": f"failed: {exc}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        if rccl_broken:   # a communicator that failed to come up may not tear down either: the line is out, leave without the destructor
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(0)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
