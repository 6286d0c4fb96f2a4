"""Does replaying te_step from a HIP graph beat enqueueing its two launches per step?  (65 536 stage03 envs)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dronechase_amd import default_config
from dronechase_amd.batched_env import BatchedEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = BatchedEnv(default_config("stage03", n_envs=N), "cuda:0"); env.reset()
acts = torch.stack([env.random_actions(7, s).clone() for s in range(64)])
static = torch.empty((N, 4), device="cuda:0")
def eager(k):
    for s in range(k): env.step(acts[s % 64], terminal=True)
def timed(fn, k):
    fn(30); torch.cuda.synchronize(); t = time.perf_counter(); fn(k); torch.cuda.synchronize(); return (time.perf_counter() - t) / k * 1e6
print(f"eager: {timed(eager, 300):.1f} us/step")
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    static.copy_(acts[0]); env.step(static, terminal=True)
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    env.step(static, terminal=True)
def graph(k):
    for s in range(k):
        static.copy_(acts[s % 64]); g.replay()
env.reset()
print(f"graph (1 step per replay, action copy per step): {timed(graph, 300):.1f} us/step")
g4 = torch.cuda.CUDAGraph()
statics = [torch.empty((N, 4), device="cuda:0") for _ in range(4)]
with torch.cuda.graph(g4):
    for q in range(4): env.step(statics[q], terminal=True)
def graph4(k):
    for s in range(0, k, 4):
        for q in range(4): statics[q].copy_(acts[(s + q) % 64])
        g4.replay()
env.reset()
print(f"graph (4 steps per replay): {timed(graph4, 300):.1f} us/step")
