"""N > 1 path on CPU: two `gloo` ranks each own a contiguous shard of the env range
(env_index_base = rank * n_local, exactly what bench.py does per GPU), step it, and the gathered
result must equal a single-process run of the whole range bit for bit (RNG is keyed on the GLOBAL env
index, so sharding is invisible).  Also exercises bench.py's barrier + max-over-ranks timing reduction.
The stepping engine here is the oracle (no GPU in this container); the HIP path is held to the same
property on one GPU by tests/test_gpu_properties.py::test_determinism_and_shard_invariance."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, steps, out_dir):
    sys.path.insert(0, ROOT)
    import time

    import torch
    import torch.distributed as dist

    from oracle import te_oracle as O

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_local = n_total // world
    cfg = O.default_config("stage03", n_envs=n_local, env_index_base=rank * n_local, seed=21)
    env = O.OracleEnv(cfg, "f32")
    env.reset()
    dist.barrier()
    t0 = time.perf_counter()
    rew, done, iner = [], [], []
    for s in range(steps):
        a = env.random_actions(1234, s)  # synthetic actions are keyed on the global env index too
        l, i, la, r, d, info = env.step(a)
        rew.append(r.copy()); done.append(d.copy()); iner.append(i.copy())
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0 + 0.01 * rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # bench.py: elapsed = max over ranks
    # host gather (north star: "host gather only, no RCCL needed")
    mine = torch.from_numpy(np.stack(rew))
    bucket = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(bucket, mine)
    if rank == 0:
        np.savez(os.path.join(out_dir, "gathered.npz"), reward=torch.cat(bucket, dim=1).numpy(), tmax=t.numpy())
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), reward=np.stack(rew), done=np.stack(done), inertial=np.stack(iner),
             tmax=t.numpy())
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_shards_equal_single_process(tmp_path):
    import torch.multiprocessing as mp

    from oracle import te_oracle as O

    n_total, steps, world = 256, 25, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, steps, str(tmp_path)), nprocs=world, join=True)
    cfg = O.default_config("stage03", n_envs=n_total, seed=21)
    env = O.OracleEnv(cfg, "f32")
    env.reset()
    rew, done, iner = [], [], []
    for s in range(steps):
        l, i, la, r, d, info = env.step(env.random_actions(1234, s))
        rew.append(r.copy()); done.append(d.copy()); iner.append(i.copy())
    rew, done, iner = np.stack(rew), np.stack(done), np.stack(iner)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    np.testing.assert_array_equal(np.concatenate([p["reward"] for p in parts], axis=1), rew)
    np.testing.assert_array_equal(np.concatenate([p["done"] for p in parts], axis=1), done)
    np.testing.assert_array_equal(np.concatenate([p["inertial"] for p in parts], axis=1), iner)
    g = np.load(tmp_path / "gathered.npz")
    np.testing.assert_array_equal(g["reward"], rew)
    assert parts[0]["tmax"] == parts[1]["tmax"] == g["tmax"]  # every rank sees the same max-over-ranks time
