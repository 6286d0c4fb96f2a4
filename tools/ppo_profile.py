import sys, time, torch
sys.path.insert(0, '/root/repo')
from dronechase_amd import default_config
from dronechase_amd.batched_env import BatchedEnv
from dronechase_amd.ppo import LidarInertialActionPolicy
N = 16384
env = BatchedEnv(default_config("stage03", n_envs=N), "cuda:0"); env.reset()
pol = LidarInertialActionPolicy().cuda()
obs = dict(zip(("lidar", "inertial_data", "last_action"), env.observe()))
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
with torch.no_grad():
    print("lidar convs     ", round(timeit(lambda: pol.lidar(obs["lidar"]))), "us")
    print("features        ", round(timeit(lambda: pol.features(obs))), "us")
    print("dist (fwd)      ", round(timeit(lambda: pol.dist(obs))), "us")
    def samp():
        d, v = pol.dist(obs); a = d.sample(); return d.log_prob(a).sum(-1)
    print("dist+sample+logp", round(timeit(samp)), "us")
    a = torch.zeros((N, 4), device="cuda:0")
    print("env.step        ", round(timeit(lambda: env.step(a, terminal=False))), "us")
    with torch.autocast("cuda", dtype=torch.bfloat16):
        print("dist bf16       ", round(timeit(lambda: pol.dist(obs))), "us")
    x = obs["lidar"].contiguous(memory_format=torch.channels_last)
    pol2 = pol.to(memory_format=torch.channels_last)
    print("convs ch_last   ", round(timeit(lambda: pol2.lidar(x))), "us")
