"""Throughput of S independent env shards per GPU, each on its own HIP stream (total env count fixed), with the
random actions of every step generated before the timed region."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dronechase_amd import default_config
from dronechase_amd.batched_env import BatchedEnv
total = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps, warm = 200, 30
dev = torch.device("cuda", 0)
for S in (1, 2, 4):
    n = total // S
    envs = [BatchedEnv(default_config("stage03", n_envs=n, env_index_base=k * n), dev) for k in range(S)]
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    acts = [torch.empty((steps + warm, n, 4), device=dev) for _ in range(S)]
    for k, e in enumerate(envs):
        e.reset()
        for i in range(steps + warm):
            e.random_actions(1234, i, out=acts[k][i])
    torch.cuda.synchronize()
    def run(i0, i1):
        for i in range(i0, i1):
            for k, e in enumerate(envs):
                with torch.cuda.stream(streams[k]):
                    e.step(acts[k][i], terminal=True)
    run(0, warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(warm, warm + steps)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"shards {S} x {n} envs: {total * steps / (t2 - t0) / 1e6:.1f} M env-steps/s, {1e6 * (t2 - t0) / steps:.1f} us per step of all shards "
          f"(host enqueue {1e6 * (t1 - t0) / steps:.1f} us)", flush=True)
    for e in envs: e.close()
    del envs, acts
