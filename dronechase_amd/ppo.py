"""On-device PPO for the batched environments (BASELINE.json config 5).

The reference trains SB3 PPO (`MultiInputPolicy` + `LidarInertialActionExtractor`,
src/core/rl_framework/agents/policies/ppo_policies.py:234-341; apps/threatengage_runner/stage03/...) against
`SubprocVecEnv`, i.e. every observation crosses process pipes and PCIe.  stable-baselines3 is not installed here, and
with 65 536 environments per GPU the rollout buffer is the thing to keep on the device: this module is a plain-PyTorch
PPO whose rollout storage, advantage estimation and minibatches never leave HBM (a 128-step rollout of 65 536 stage03
envs is 34 GB of observations — 288 GB per MI355X is what makes that layout possible).  PyTorch-ROCm is used for
autograd and the dense layers (rocBLAS/MIOpen): the environment side stays the HIP kernels of libthreatengage.so.

  * topology = the reference's extractor: LIDAR conv(k4,s4,32) -> conv(k2,s2,64) -> flatten; inertial and
    last_action 3 x Linear(128); concat -> Linear(256); then SB3's default pi / vf heads (2 x 64, tanh) and a
    state-independent log-std (stable_baselines3 ActorCriticPolicy defaults);
  * losses / GAE = the PPO of Schulman et al. 2017 with SB3's defaults (clip 0.2, gae_lambda 0.95, gamma 0.99,
    vf_coef 0.5, ent_coef 0, max_grad_norm 0.5, advantage normalisation per minibatch, 10 epochs);
  * multi-GPU: one process per GPU, each with its own env shard; gradients are averaged with
    torch.distributed all_reduce (RCCL) — the only collective of the whole system, once per minibatch.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.nn as nn


class LidarInertialActionPolicy(nn.Module):
    """Actor-critic over {"lidar" [B,C,13,26], "inertial_data" [B,15], "last_action" [B,4]}."""

    def __init__(self, lidar_shape=(3, 13, 26), inertial_dim: int = 15, action_dim: int = 4, features_dim: int = 256):
        super().__init__()
        c = lidar_shape[0]
        self.lidar = nn.Sequential(nn.Conv2d(c, 32, kernel_size=4, stride=4), nn.ReLU(),
                                   nn.Conv2d(32, 64, kernel_size=2, stride=2), nn.ReLU(), nn.Flatten())
        with torch.no_grad():
            n_lidar = self.lidar(torch.zeros(1, *lidar_shape)).shape[1]

        def mlp(n_in):
            return nn.Sequential(nn.Linear(n_in, 128), nn.ReLU(), nn.Linear(128, 128), nn.ReLU(), nn.Linear(128, 128), nn.ReLU())

        self.inertial, self.action = mlp(inertial_dim), mlp(action_dim)
        self.final = nn.Sequential(nn.Linear(n_lidar + 256, features_dim), nn.ReLU())
        self.pi = nn.Sequential(nn.Linear(features_dim, 64), nn.Tanh(), nn.Linear(64, 64), nn.Tanh())
        self.vf = nn.Sequential(nn.Linear(features_dim, 64), nn.Tanh(), nn.Linear(64, 64), nn.Tanh())
        self.mu = nn.Linear(64, action_dim)
        self.value = nn.Linear(64, 1)
        self.log_std = nn.Parameter(torch.zeros(action_dim))

    def features(self, obs: Dict[str, torch.Tensor]) -> torch.Tensor:
        z = torch.cat((self.lidar(obs["lidar"]), self.inertial(obs["inertial_data"]), self.action(obs["last_action"])), dim=1)
        return self.final(z)

    def forward(self, obs):
        f = self.features(obs)
        return self.mu(self.pi(f)), self.value(self.vf(f)).squeeze(-1)

    def dist(self, obs):
        mu, v = self(obs)
        # validate_args=False: the argument check is a host synchronisation per call (and cannot be captured in a graph)
        return torch.distributions.Normal(mu, self.log_std.exp().expand_as(mu), validate_args=False), v


class PolicyDriver:
    """SB3-style `predict` over a LidarInertialActionPolicy, on the device: what ThreatEngageVecEnv.update_model (exp05:
    the ally flown by a copy of the learning policy, apps/threatengage_runner/stage03/experiments/05/
    bo_exp05_vFinal_home_office_app.py:140,179) takes when the observations should not leave HBM."""
    accepts_torch = True

    def __init__(self, policy: nn.Module):
        self.policy = policy
        self.low = None

    @torch.no_grad()
    def predict(self, observation: Dict[str, torch.Tensor], state=None, episode_start=None, deterministic: bool = True):
        dist, _ = self.policy.dist(observation)
        a = dist.mean if deterministic else dist.sample()
        if self.low is None:
            self.low = torch.tensor([-1.0, -1.0, -1.0, 0.0], device=a.device)
        return torch.max(torch.min(a, torch.ones_like(a)), self.low), None


@dataclass
class PPOConfig:
    n_steps: int = 128
    batch_size: int = 2048
    n_epochs: int = 10
    gamma: float = 0.99
    gae_lambda: float = 0.95
    clip_range: float = 0.2
    vf_coef: float = 0.5
    ent_coef: float = 0.0
    max_grad_norm: float = 0.5
    learning_rate: float = 3e-4
    use_graph: bool = True       # collect(): policy forward + sampling + te_step captured once in a HIP graph and replayed per step
    # update(): fused Adam (one multi-tensor kernel instead of ~25 x 6 element-wise launches per minibatch) and bf16 autocast of the forward /
    # backward of the policy (fp32 master weights, fp32 losses and optimiser): measured + 16 % on the update at 32 768-sample minibatches
    # (tools/ppo_update_profile.py).  Off by default: fp32 / plain Adam is what SB3 runs.
    fast_learner: bool = False
    reward_scale: float = 1e-3   # rewards reach +-1000 (exp03_vFinal_task.py:423-515); SB3 users wrap VecNormalize


class RolloutBuffer:
    """[n_steps, N, ...] tensors on the environment's device."""

    def __init__(self, n_steps: int, n_envs: int, obs_shapes: Dict[str, tuple], device):
        f = dict(dtype=torch.float32, device=device)
        self.obs = {k: torch.empty((n_steps, n_envs, *s), **f) for k, s in obs_shapes.items()}
        self.actions = torch.empty((n_steps, n_envs, 4), **f)
        self.logp = torch.empty((n_steps, n_envs), **f)
        self.values = torch.empty((n_steps, n_envs), **f)
        self.rewards = torch.empty((n_steps, n_envs), **f)
        self.dones = torch.empty((n_steps, n_envs), **f)          # done AFTER this step
        self.adv = torch.empty((n_steps, n_envs), **f)
        self.ret = torch.empty((n_steps, n_envs), **f)

    def bytes(self) -> int:
        ts = list(self.obs.values()) + [self.actions, self.logp, self.values, self.rewards, self.dones, self.adv, self.ret]
        return sum(t.numel() * t.element_size() for t in ts)

    @torch.no_grad()
    def finish(self, last_value: torch.Tensor, gamma: float, lam: float) -> None:
        """GAE(lambda).  An env that auto-reset at step t starts a new episode at t+1: no bootstrap across it
        (terminations only: the reference never truncates, exp03_vFinal_environment.py:166)."""
        gae = torch.zeros_like(last_value)
        nxt = last_value
        for t in reversed(range(self.rewards.shape[0])):
            nonterminal = 1.0 - self.dones[t]
            delta = self.rewards[t] + gamma * nxt * nonterminal - self.values[t]
            gae = delta + gamma * lam * nonterminal * gae
            self.adv[t] = gae
            nxt = self.values[t]
        self.ret.copy_(self.adv + self.values)


class PPO:
    """`env` is a dronechase_amd.batched_env.BatchedEnv (classic own-sphere observation)."""

    def __init__(self, env, cfg: Optional[PPOConfig] = None, policy: Optional[nn.Module] = None, seed: int = 0):
        self.env, self.cfg = env, cfg or PPOConfig()
        self.device = env.device
        ecfg = getattr(env, "cfg", None)
        if ecfg is not None and (int(ecfg.ally_policy) == 3 or (int(ecfg.evaluation) >> 8) != 0):
            # exp05 / Evaluation_Task "nn" drivers: somebody has to answer te_observe_wingman with te_set_wingman_actions
            # before every te_step; this rollout loop does not, and the wingman would fly a stale set-point
            raise ValueError("PPO drives the agent only: an environment with caller-driven wingmen (exp05, evaluation driver mask) "
                             "needs its wingman driver in the loop (ThreatEngageVecEnv.update_model), not this rollout")
        torch.manual_seed(seed)
        self.policy = (policy or LidarInertialActionPolicy(lidar_shape=tuple(env.lidar.shape[1:]))).to(self.device)
        fused = bool(self.cfg.fast_learner) and self.device.type == "cuda"
        self.opt = torch.optim.Adam(self.policy.parameters(), lr=self.cfg.learning_rate, eps=1e-5, **({"fused": True} if fused else {}))
        shapes = {"lidar": tuple(env.lidar.shape[1:]), "inertial_data": (env.inertial.shape[1],), "last_action": (4,)}
        self.buf = RolloutBuffer(self.cfg.n_steps, env.N, shapes, self.device)
        self.low = torch.tensor([-1.0, -1.0, -1.0, 0.0], device=self.device)
        self.high = torch.ones(4, device=self.device)
        self.distributed = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
        if self.distributed:  # same initial weights on every rank
            for p in self.policy.parameters():
                torch.distributed.broadcast(p.data, src=0)
        # ONE gradient bucket: every parameter's .grad is a view of this flat buffer, so data-parallel training averages the
        # gradients of a minibatch with a single all-reduce (RCCL over xGMI: a ring is per-link bound, 25 small collectives per
        # minibatch would be latency-bound; the policy has ~0.3 M parameters = 1.2 MB, one bucket)
        params = [p for p in self.policy.parameters() if p.requires_grad]
        self._flat_grad = torch.zeros(sum(p.numel() for p in params), dtype=params[0].dtype, device=self.device)
        off = 0
        for p in params:
            p.grad = self._flat_grad[off:off + p.numel()].view_as(p)
            off += p.numel()
        env.reset()
        self.direct = env.N % 2 == 0   # slot t of the LIDAR buffer starts on a 16-byte boundary (4 056 bytes per env)
        self._obs = None               # set by the first collect(): te_observe of the reset state
        self.num_timesteps = 0

    def _slot(self, t: int):
        b = self.buf
        return b.obs["lidar"][t], b.obs["inertial_data"][t], b.obs["last_action"][t]

    # ------------------------------------------------------------------ graph-captured rollout step
    def _capture_step(self) -> None:
        """One rollout step = ~40 small PyTorch launches (policy forward, sampling, clamps) + the two env kernels; at
        16 384 envs that is launch-bound in eager mode (1.9 ms per step, of which the kernels need ~0.8).  The step is
        captured once in a HIP graph on static tensors and replayed: te_step enqueues on PyTorch's current stream, so
        its launches are captured like any other."""
        b = self.buf
        self._g_obs = {k: torch.empty_like(v[0]) for k, v in b.obs.items()}
        self._g = dict(a=torch.empty_like(b.actions[0]), logp=torch.empty_like(b.logp[0]), v=torch.empty_like(b.values[0]),
                       reward=torch.empty_like(b.rewards[0]), done=torch.empty_like(b.dones[0]))
        for k in self._g_obs:
            self._g_obs[k].copy_(self._obs[k])

        def step_once():
            # the diagonal Gaussian by hand: torch.normal on expanded tensors checks its arguments on the host, which
            # a capturing stream does not permit.  log N(a; mu, sigma) = -(a - mu)^2 / (2 sigma^2) - log sigma - log sqrt(2 pi)
            mu, v = self.policy(self._g_obs)
            log_std = self.policy.log_std
            eps = torch.randn_like(mu)
            a = mu + log_std.exp() * eps
            logp = (-0.5 * eps * eps - log_std - 0.9189385332046727).sum(-1)
            self._g["a"].copy_(a); self._g["logp"].copy_(logp); self._g["v"].copy_(v)
            lidar, inertial, last_action, reward, done, _ = self.env.step(torch.max(torch.min(a, self.high), self.low).contiguous(), terminal=False)
            self._g["reward"].copy_(reward); self._g["done"].copy_(done)
            self._g_obs["lidar"].copy_(lidar); self._g_obs["inertial_data"].copy_(inertial); self._g_obs["last_action"].copy_(last_action)

        state = self.env.get_state().clone()          # warm-up and capture must not advance the environments
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(2):
                step_once()
        torch.cuda.current_stream(self.device).wait_stream(side)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            step_once()
        self.env.set_state(state)
        for k in self._g_obs:
            self._g_obs[k].copy_(self._obs[k])

    @torch.no_grad()
    def _collect_graph(self) -> Dict[str, float]:
        b, c = self.buf, self.cfg
        if self._obs is None:
            self._obs = dict(zip(("lidar", "inertial_data", "last_action"), self.env.observe()))
        if not hasattr(self, "_graph"):
            self._capture_step()
        ep_rew = torch.zeros((), device=self.device); ep_n = torch.zeros((), dtype=torch.int64, device=self.device)
        for t in range(c.n_steps):
            for k in b.obs:
                b.obs[k][t].copy_(self._g_obs[k])
            self._graph.replay()
            b.actions[t].copy_(self._g["a"]); b.logp[t].copy_(self._g["logp"]); b.values[t].copy_(self._g["v"])
            torch.mul(self._g["reward"], c.reward_scale, out=b.rewards[t]); b.dones[t].copy_(self._g["done"])
            ep_rew += self._g["reward"].mean(); ep_n += self._g["done"].sum().long()
        self._obs = self._g_obs
        _, last_v = self.policy(self._obs)
        b.finish(last_v, c.gamma, c.gae_lambda)
        self.num_timesteps += c.n_steps * self.env.N
        return {"mean_step_reward": float(ep_rew) / c.n_steps, "episodes_finished": int(ep_n)}

    @torch.no_grad()
    def collect(self) -> Dict[str, float]:
        if self.cfg.use_graph and self.device.type == "cuda":
            return self._collect_graph()
        return self._collect_eager()

    @torch.no_grad()
    def _collect_eager(self) -> Dict[str, float]:
        """The environment writes the observation of step t+1 straight into slot t+1 of the rollout buffer (te_step takes
        the destination pointers): no per-step copy of the 4 KB/env observation, and no host synchronisation inside the
        loop.  The observation after the last step goes to the env's own buffers and seeds slot 0 of the next rollout."""
        b, c = self.buf, self.cfg
        ep_rew = torch.zeros((), device=self.device); ep_n = torch.zeros((), dtype=torch.int64, device=self.device)
        if self._obs is None:
            self._obs = dict(zip(("lidar", "inertial_data", "last_action"), self.env.observe()))
        for k in b.obs:
            b.obs[k][0].copy_(self._obs[k])
        for t in range(c.n_steps):
            dist, v = self.policy.dist({k: o[t] for k, o in b.obs.items()})
            a = dist.sample()
            b.actions[t], b.logp[t], b.values[t] = a, dist.log_prob(a).sum(-1), v
            dest = self._slot(t + 1) if (self.direct and t + 1 < c.n_steps) else None
            lidar, inertial, last_action, reward, done, _info = self.env.step(torch.max(torch.min(a, self.high), self.low).contiguous(),
                                                                              terminal=False, out=dest)
            if dest is None and t + 1 < c.n_steps:   # odd n_envs: the slots are not 16-byte aligned, copy instead
                for k, src in zip(("lidar", "inertial_data", "last_action"), (lidar, inertial, last_action)):
                    b.obs[k][t + 1].copy_(src)
            b.rewards[t] = reward * c.reward_scale
            b.dones[t] = done.float()
            ep_rew += reward.mean(); ep_n += done.sum()   # stays on the device: one sync per rollout, not per step
        self._obs = {"lidar": lidar, "inertial_data": inertial, "last_action": last_action}
        _, last_v = self.policy(self._obs)
        ep_rew, ep_n = float(ep_rew), int(ep_n)
        b.finish(last_v, c.gamma, c.gae_lambda)
        self.num_timesteps += c.n_steps * self.env.N
        return {"mean_step_reward": ep_rew / c.n_steps, "episodes_finished": ep_n}

    def update(self) -> Dict[str, float]:
        b, c = self.buf, self.cfg
        T, N = b.rewards.shape
        flat = lambda x: x.reshape(T * N, *x.shape[2:])
        obs = {k: flat(v) for k, v in b.obs.items()}
        actions, old_logp, adv, ret = flat(b.actions), flat(b.logp), flat(b.adv), flat(b.ret)
        # running sums stay on the device: one host read per update(), not four per minibatch (each float() drains the stream)
        acc = torch.zeros(4, device=self.device)
        n_batches = 0
        amp = bool(c.fast_learner) and self.device.type == "cuda"
        for _ in range(c.n_epochs):
            perm = torch.randperm(T * N, device=self.device)
            for s in range(0, T * N, c.batch_size):
                idx = perm[s:s + c.batch_size]
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                    mu, v = self.policy({k: o[idx] for k, o in obs.items()})
                mu, v = mu.float(), v.float()
                dist = torch.distributions.Normal(mu, self.policy.log_std.exp().expand_as(mu), validate_args=False)
                logp = dist.log_prob(actions[idx]).sum(-1)
                a = adv[idx]
                a = (a - a.mean()) / (a.std() + 1e-8)
                ratio = (logp - old_logp[idx]).exp()
                pg = -torch.min(a * ratio, a * ratio.clamp(1 - c.clip_range, 1 + c.clip_range)).mean()
                vl = torch.nn.functional.mse_loss(v, ret[idx])
                ent = dist.entropy().sum(-1).mean()
                loss = pg + c.vf_coef * vl - c.ent_coef * ent
                self._flat_grad.zero_()          # the parameters' .grad are views of it: backward accumulates in place
                loss.backward()
                if self.distributed:             # the system's only collective: mean gradient over the ranks, one bucket
                    torch.distributed.all_reduce(self._flat_grad)
                    self._flat_grad.div_(torch.distributed.get_world_size())
                nn.utils.clip_grad_norm_(self.policy.parameters(), c.max_grad_norm)
                self.opt.step()
                with torch.no_grad():
                    acc += torch.stack((pg.detach(), vl.detach(), ent.detach(), ((ratio - 1).abs() > c.clip_range).float().mean()))
                n_batches += 1
        pg_s, vl_s, ent_s, clip_s = (acc / max(n_batches, 1)).tolist()
        return {"pg_loss": pg_s, "v_loss": vl_s, "entropy": ent_s, "clip_frac": clip_s}

    def learn(self, total_timesteps: int, log=None):
        while self.num_timesteps < total_timesteps:
            r = self.collect()
            u = self.update()
            if log:
                log({**r, **u, "timesteps": self.num_timesteps})
        return self
