#!/usr/bin/env python3
"""Where PPO.update() spends a minibatch at 16 384 envs (batch 32 768), and what cheap PyTorch-side variants buy:
the LIDAR extractor's two stride = kernel convolutions as unfold + GEMM, bf16 autocast, a fused Adam."""
import os, sys, time
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn, torch.nn.functional as F
from dronechase_amd.ppo import LidarInertialActionPolicy

B = 32768
dev = "cuda:0"
obs = {"lidar": torch.rand((B, 3, 13, 26), device=dev), "inertial_data": torch.rand((B, 15), device=dev), "last_action": torch.rand((B, 4), device=dev)}
act = torch.rand((B, 4), device=dev); adv = torch.randn(B, device=dev); ret = torch.randn(B, device=dev); old = torch.randn(B, device=dev)


class PatchLidar(nn.Module):
    """The two convolutions of LidarInertialActionExtractor have stride = kernel: each is one GEMM over non-overlapping patches."""
    def __init__(self, seq):
        super().__init__()
        self.c1, self.c2 = seq[0], seq[2]
    def forward(self, x):
        Bn = x.shape[0]
        p = x[:, :, :12, :24].reshape(Bn, 3, 3, 4, 6, 4).permute(0, 2, 4, 1, 3, 5).reshape(Bn * 18, 48)
        y = F.relu(F.linear(p, self.c1.weight.view(32, 48), self.c1.bias)).view(Bn, 3, 6, 32)
        q = y[:, :2].reshape(Bn, 2, 3, 2, 32).permute(0, 2, 4, 1, 3).reshape(Bn * 3, 128)
        z = F.relu(F.linear(q, self.c2.weight.view(64, 128), self.c2.bias)).view(Bn, 3, 64)
        return z.permute(0, 2, 1).reshape(Bn, 192)


def loss_of(pol):
    d, v = pol.dist(obs)
    logp = d.log_prob(act).sum(-1)
    ratio = (logp - old).exp()
    pg = -torch.min(adv * ratio, adv * ratio.clamp(0.8, 1.2)).mean()
    return pg + 0.5 * F.mse_loss(v, ret)


def bench(name, pol, opt, autocast=False, n=20):
    def step():
        opt.zero_grad(set_to_none=False)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            l = loss_of(pol)
        l.backward()
        nn.utils.clip_grad_norm_(pol.parameters(), 0.5)
        opt.step()
    for _ in range(5): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    print(f"{name:46s} {dt * 1e3:7.2f} ms per minibatch of {B} = {B / dt / 1e6:6.1f} M samples/s", flush=True)


torch.manual_seed(0)
base = LidarInertialActionPolicy().to(dev)
x = obs["lidar"][:64]
pl = PatchLidar(base.lidar)
print("patch extractor == conv extractor:", float((pl(x) - base.lidar(x)).abs().max()))
bench("baseline (Conv2d via MIOpen, Adam)", base, torch.optim.Adam(base.parameters(), lr=3e-4, eps=1e-5))
p2 = LidarInertialActionPolicy().to(dev); p2.lidar = PatchLidar(p2.lidar)
bench("convolutions as unfold + GEMM", p2, torch.optim.Adam(p2.parameters(), lr=3e-4, eps=1e-5))
p3 = LidarInertialActionPolicy().to(dev); p3.lidar = PatchLidar(p3.lidar)
bench("  + fused Adam", p3, torch.optim.Adam(p3.parameters(), lr=3e-4, eps=1e-5, fused=True))
p4 = LidarInertialActionPolicy().to(dev); p4.lidar = PatchLidar(p4.lidar)
bench("  + fused Adam + bf16 autocast", p4, torch.optim.Adam(p4.parameters(), lr=3e-4, eps=1e-5, fused=True), autocast=True)
p5 = LidarInertialActionPolicy().to(dev)
bench("Conv2d + fused Adam + bf16 autocast", p5, torch.optim.Adam(p5.parameters(), lr=3e-4, eps=1e-5, fused=True), autocast=True)
