"""N > 1 path of the PRODUCT on the one-GPU box: two `gloo` ranks share cuda:0, each owns a contiguous shard of the env range
(env_index_base = rank * n_local: what bench.py does per GPU) in its own te_env, steps it through the C ABI, the results are
gathered on the host (north star: "host gather only, no RCCL needed") and must equal one te_env stepping the whole range bit for
bit.  Then bench.py itself as the driver launches it for N = 2 (torch.distributed.run, TE_BENCH_BACKEND=gloo because both ranks
share one device; the real runs use nccl = RCCL, one rank per GPU): the strong-scaling headline BASELINE.json names (total envs
fixed) and the weak-scaling block.  The CPU twin of the first test (the oracle as the stepping engine) is tests/test_sharding_gloo.py."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


WORKER = r"""
import os, sys, time
import numpy as np
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from dronechase_amd import default_config
from dronechase_amd.batched_env import BatchedEnv
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
n_total, steps, out_dir = {n_total}, {steps}, {out_dir!r}
n_local = n_total // world
env = BatchedEnv(default_config("stage03", n_envs=n_local, env_index_base=rank * n_local, seed=21), "cuda:0")
env.reset()
dist.barrier(); torch.cuda.synchronize(); t0 = time.perf_counter()
rew, done, iner, lid = [], [], [], []
for s in range(steps):
    l, i, la, r, d, info = env.step(env.random_actions(1234, s))
    rew.append(r.cpu()); done.append(d.cpu()); iner.append(i.cpu()); lid.append(l.sum(dim=(1, 2, 3)).cpu())
torch.cuda.synchronize(); dist.barrier()
t = torch.tensor([time.perf_counter() - t0 + 0.01 * rank], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
mine = torch.stack(rew)
bucket = [torch.empty_like(mine) for _ in range(world)]
dist.all_gather(bucket, mine)                                   # the learner-side host gather
if rank == 0:
    np.savez(os.path.join(out_dir, "gathered.npz"), reward=torch.cat(bucket, dim=1).numpy(), tmax=t.numpy())
np.savez(os.path.join(out_dir, f"rank{{rank}}.npz"), reward=mine.numpy(), done=torch.stack(done).numpy(), inertial=torch.stack(iner).numpy(),
         lidar_sum=torch.stack(lid).numpy(), state=env.get_state().cpu().numpy(), tmax=t.numpy())
env.close(); dist.destroy_process_group()
"""


def _launch(nproc, script_args, env_extra, timeout):
    env = dict(os.environ, **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), *script_args]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_two_product_ranks_equal_one_te_env(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    from dronechase_amd import default_config
    from dronechase_amd.batched_env import BatchedEnv
    n_total, steps = 1024, 40
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, n_total=n_total, steps=steps, out_dir=str(tmp_path)))
    out = _launch(2, [str(script)], {}, 600)
    assert out.returncode == 0, out.stderr[-3000:]
    env = BatchedEnv(default_config("stage03", n_envs=n_total, seed=21), "cuda:0")
    env.reset()
    rew, done, iner, lid = [], [], [], []
    for s in range(steps):
        l, i, la, r, d, info = env.step(env.random_actions(1234, s))
        rew.append(r.cpu().numpy().copy()); done.append(d.cpu().numpy().copy()); iner.append(i.cpu().numpy().copy()); lid.append(l.sum(dim=(1, 2, 3)).cpu().numpy())
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(2)]
    np.testing.assert_array_equal(np.concatenate([p["reward"] for p in parts], axis=1), np.stack(rew))
    np.testing.assert_array_equal(np.concatenate([p["done"] for p in parts], axis=1), np.stack(done))
    np.testing.assert_array_equal(np.concatenate([p["inertial"] for p in parts], axis=1), np.stack(iner))
    np.testing.assert_array_equal(np.concatenate([p["lidar_sum"] for p in parts], axis=1), np.stack(lid))
    g = np.load(tmp_path / "gathered.npz")
    np.testing.assert_array_equal(g["reward"], np.stack(rew))
    assert parts[0]["tmax"] == parts[1]["tmax"] == g["tmax"]     # every rank sees the same max-over-ranks time
    # the state blobs of the shards are the halves of the whole env's blob (drone records, then env records)
    from dronechase_amd import config as K
    whole = env.get_state().cpu().numpy()
    D, n = 11, n_total // 2
    dr = whole[: n_total * D * K.DRONE_WORDS].reshape(n_total, -1); er = whole[n_total * D * K.DRONE_WORDS:].reshape(n_total, -1)
    for r, p in enumerate(parts):
        st = p["state"]
        np.testing.assert_array_equal(st[: n * D * K.DRONE_WORDS].reshape(n, -1), dr[r * n:(r + 1) * n])
        np.testing.assert_array_equal(st[n * D * K.DRONE_WORDS:].reshape(n, -1), er[r * n:(r + 1) * n])
    env.close()


def test_bench_two_ranks_reports_the_strong_scaling_point():
    """`bench.py --gpus 2` as the driver launches it: the headline is the metric's configuration (--total-envs split over the
    ranks, "scaling": "strong"), the 65 536-envs-per-GPU figure travels in `weak_scaling` (here: scaled down to fit two ranks on one card)."""
    out = _launch(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3", "--total-envs", "8192", "--headline-only",
                      "--no-cpu-baseline"], {"TE_BENCH_BACKEND": "gloo"}, 900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                       # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 10 and d["warmup"] == 3
    assert d["config"]["total_envs"] == 8192 and d["config"]["envs_per_gpu"] == 4096
    assert abs(d["value"] - 8192 * 10 / (d["ms_per_step"] * 1e-3 * 10)) / d["value"] < 1e-6      # whole-job envs over the max-over-ranks time
    w = d["weak_scaling"]
    assert w["scaling"] == "weak" and w["config"]["envs_per_gpu"] == 8192 and w["config"]["total_envs"] == 16384
    assert abs(w["value"] - 16384 / (w["ms_per_step"] * 1e-3)) / w["value"] < 1e-6
    assert "cpu_baseline" not in d                               # rank 0 at N = 1 only


def test_plain_bench_command_with_two_gpus_launches_its_own_ranks():
    """`python3 bench.py --gpus 2 ...` with no WORLD_SIZE in the environment (the shape of the N = 1 command): bench.py starts
    torch.distributed.run itself as a fresh child before touching the GPU, relays rank 0's single JSON line and exits with the child's code."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["TE_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3", "--total-envs", "8192",
                          "--headline-only", "--no-cpu-baseline", "--no-weak-block"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["envs_per_gpu"] == 4096 and "weak_scaling" not in d
    assert d["config"]["control_plane"].startswith("gloo")
    # a failing child must fail the plain command too (WORLD_SIZE would not divide the envs)
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--total-envs", "8191",
                          "--headline-only", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert bad.returncode != 0 and not [l for l in bad.stdout.splitlines() if l.startswith("{")]


def test_ppo_example_with_two_ranks_sharing_the_gpu():
    """BASELINE config 5 rehearsed with N = 2: examples/ppo_stage03.py under torch.distributed.run, two ranks on the one GPU
    (TE_PPO_BACKEND=gloo), each with its own env shard and rollout, ONE gradient all-reduce per minibatch; the replicas must end identical."""
    out = _launch(2, [os.path.join(ROOT, "examples", "ppo_stage03.py"), "--envs", "512", "--iters", "2", "--n-steps", "8", "--batch-size", "1024",
                      "--epochs", "2"], {"TE_PPO_BACKEND": "gloo"}, 900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    iters = [l for l in lines if "iter" in l]
    assert [l["iter"] for l in iters] == [0, 1] and all(l["env_steps"] == 8 * 512 * 2 for l in iters)      # whole-job env-steps: both shards
    assert all(np.isfinite(l["pg_loss"]) and np.isfinite(l["v_loss"]) for l in iters)
    sync = [l for l in lines if "replicas_in_sync" in l]
    assert len(sync) == 1 and sync[0]["world"] == 2 and sync[0]["replicas_in_sync"] is True and sync[0]["param_abs_sum"] > 0
