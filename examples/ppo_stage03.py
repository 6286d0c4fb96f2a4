#!/usr/bin/env python3
"""stage03 + PPO with everything on the GPU (BASELINE.json config 5): env shard, rollout storage, policy, update.

    python examples/ppo_stage03.py --envs 65536 --iters 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/ppo_stage03.py --envs 8192

One process per GPU; each rank owns `--envs` environments (RNG keyed on the global env index) and the gradient
all-reduce is the only collective.  Mirrors apps/threatengage_runner/stage03 (PPO + LidarInertialActionExtractor,
rl_frequency 15, dome 20) without stable-baselines3."""
import argparse, json, os, sys, time

os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")  # the exhaustive convolution search costs minutes per new batch shape

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=8192)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--n-steps", type=int, default=128)
    ap.add_argument("--batch-size", type=int, default=32768)
    ap.add_argument("--epochs", type=int, default=4)
    ap.add_argument("--task", default="stage03")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from dronechase_amd import default_config
    from dronechase_amd.batched_env import BatchedEnv
    from dronechase_amd.ppo import PPO, PPOConfig

    world, rank, local = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    # TE_PPO_BACKEND=gloo: rehearsal on a box with fewer GPUs than ranks (the ranks share devices round-robin, the gradient bucket crosses
    # through the host); the real thing is nccl (= RCCL), one rank per GPU
    backend = os.environ.get("TE_PPO_BACKEND", "nccl")
    local = local if backend == "nccl" else local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    env = BatchedEnv(default_config(args.task, n_envs=args.envs, env_index_base=rank * args.envs), dev)
    ppo = PPO(env, PPOConfig(n_steps=args.n_steps, batch_size=args.batch_size, n_epochs=args.epochs), seed=0)
    if rank == 0:
        print(f"rollout buffer {ppo.buf.bytes() / 2**30:.1f} GiB on {dev}; policy parameters {sum(p.numel() for p in ppo.policy.parameters())}", flush=True)
    for it in range(args.iters):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = ppo.collect()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        u = ppo.update()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        if rank == 0:
            n = args.n_steps * args.envs * world
            print(json.dumps({"iter": it, "env_steps": n, "collect_env_steps_per_s": n / (t1 - t0), "train_env_steps_per_s": n / (t2 - t0),
                              **{k: round(v, 5) for k, v in {**r, **u}.items()}}), flush=True)
    if world > 1:   # every rank applied the same mean gradient to the same initial weights: the replicas must still be identical
        flat = torch.cat([p.detach().reshape(-1) for p in ppo.policy.parameters()]).double()
        lo, hi = flat.clone(), flat.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"world": world, "replicas_in_sync": bool(torch.equal(lo, hi)), "param_abs_sum": float(flat.abs().sum())}), flush=True)
    env.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
