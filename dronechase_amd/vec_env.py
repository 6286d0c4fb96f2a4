"""ThreatEngageVecEnv — the drop-in for the object `ReinforcementLearningPipeline.create_vectorized_environment`
returns (src/core/rl_framework/utils/pipeline.py:31-61): the SB3 2.6 `VecEnv` surface over ONE batched GPU
environment instead of N SubprocVecEnv workers.

SB3 is not required: the class is duck-typed, and additionally derives from
stable_baselines3.common.vec_env.VecEnv when that package is importable so isinstance checks pass."""
from __future__ import annotations

from typing import Any, Dict, Iterable, List, Optional, Sequence, Union

import numpy as np

from . import config as K
from . import spaces
from ._lib import default_config

try:  # pragma: no cover - depends on the box
    from stable_baselines3.common.vec_env.base_vec_env import VecEnv as _SB3VecEnv  # type: ignore
except Exception:
    _SB3VecEnv = object

INFO_KEYS = ("agent_kills", "allies_kills", "deads", "current_wave")  # exp03_vFinal_task.py:571-578


def make_config(task: str, num_envs: int, dome_radius: Optional[float] = None, rl_frequency: int = 15, seed: int = 0,
                env_index_base: int = 0, **overrides) -> K.Config:
    """Reference constructor kwargs (dome_radius, rl_frequency, GUI; pipeline.py:45-51) -> te_config."""
    cfg = default_config(task, n_envs=num_envs, seed=seed, env_index_base=env_index_base)
    if dome_radius is not None:
        cfg.dome_radius = float(dome_radius)
        # lidar radius follows the dome only where the reference ties them (stages.py:397-401,
        # exp03_vFinal_task.py:641-643); stage01 hard-codes 20 (level2/components/quadcopter_manager.py:44)
        if cfg.task != K.TASK_STAGE01:
            cfg.lidar_radius = 2.0 * float(dome_radius)
    if rl_frequency != 15:
        agg = int(120 / rl_frequency)  # frequency_adjustments (exp03_vFinal_environment.py:107-110)
        if agg < 1:
            raise ValueError("rl_frequency must be <= 120")
        cfg.substeps = 2 * agg
        if cfg.task == K.TASK_STAGE01:
            cfg.max_step = 20 * rl_frequency  # pyflyt_level2_environment_modified_v2.py:47
    return K.apply_overrides(cfg, **overrides)


class LazyInfos(Sequence):
    """infos without materialising N dicts per step: dicts are built on access.  `info` / `done` may be zero-argument callables
    (output="torch": nothing crosses PCIe, and the stream is not drained, unless somebody actually reads an info)."""

    def __init__(self, info, done, terminal, still_valid=None):
        self._info, self._done, self._terminal = info, done, terminal
        self._n = None if callable(done) else len(done)
        # output="torch": the terminal observations are VIEWS of the backend's terminal buffers, which hold the rows of the step that produced
        # these infos and are rewritten by the next one.  `still_valid()` turns a late read into an error instead of another step's rows.
        self._still_valid = still_valid

    def _host(self):
        if callable(self._info):
            self._info, self._done = self._info(), self._done()
            if callable(self._terminal):
                self._terminal = self._terminal(self._done)
        return self._info, self._done

    def __len__(self):
        return len(self._host()[1]) if self._n is None else self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        info, done = self._host()
        r = info[i]
        d = {"agent_kills": int(r[0]), "allies_kills": int(r[1]), "deads": int(r[2]), "current_wave": int(r[3]),
             "TimeLimit.truncated": False}  # the reference always returns truncated=False (exp03_vFinal_environment.py:167)
        if done[i] and self._terminal is not None:
            if self._still_valid is not None and not self._still_valid():
                raise RuntimeError("infos[i]['terminal_observation'] was read after a later step(): with output='torch' the terminal observations are "
                                   "views of device buffers that hold the LAST step's rows only (counters, done and the key itself stay valid). "
                                   "Read them before the next step, or clone the rows you keep.")
            d["terminal_observation"] = {k: v[i] for k, v in self._terminal.items()}
        return d

    def has_terminal_observation(self, i) -> bool:
        """Whether env i finished an episode in the step these infos belong to (valid for ever, unlike the rows themselves)."""
        return bool(self._host()[1][i]) and self._terminal is not None

    def materialise(self) -> List[Dict[str, Any]]:
        """The list of N dicts SB3 iterates over (infos="dicts"): one C-level conversion of the info rows, then a dict display per env."""
        import gc

        info, done = self._host()
        a, b, c, w = info.T.tolist()          # four lists of Python ints in one C-level pass each
        on = gc.isenabled()
        gc.disable()                          # N fresh containers would trigger a generation-2 pass over the previous steps' dicts every few hundred
        try:
            out = [{"agent_kills": x, "allies_kills": y, "deads": z, "current_wave": r, "TimeLimit.truncated": False} for x, y, z, r in zip(a, b, c, w)]
        finally:
            if on:
                gc.enable()
        if self._terminal is not None:
            for i in np.flatnonzero(done):
                out[i]["terminal_observation"] = {k: v[int(i)] for k, v in self._terminal.items()}
        return out


class ThreatEngageVecEnv(_SB3VecEnv):  # type: ignore[misc]
    """SB3 VecEnv API: reset / step_async / step_wait / step / close / seed / get_attr / set_attr /
    env_method / env_is_wrapped, attributes num_envs, observation_space, action_space, reset_infos,
    render_mode.  Auto-reset with infos[i]["terminal_observation"] as SB3 expects."""

    def __init__(self, task: str = "stage03", num_envs: int = 1024, device: str = "cuda:0",
                 dome_radius: Optional[float] = None, rl_frequency: int = 15, GUI: bool = False, seed: int = 0,
                 env_index_base: int = 0, output: str = "numpy", infos: str = "dicts", backend=None,
                 persistent_obs: Optional[bool] = None, **overrides):
        if GUI:
            raise ValueError("GUI=True has no batched equivalent (the reference forces n_envs=1 with a PyBullet window)")
        if output not in ("numpy", "torch") or infos not in ("dicts", "lazy"):
            raise ValueError("output in {'numpy','torch'}, infos in {'dicts','lazy'}")
        self.task = task
        self.cfg = make_config(task, num_envs, dome_radius, rl_frequency, seed, env_index_base, auto_reset=1, **overrides)
        self.output, self.infos_mode = output, infos
        if backend is None:
            from .batched_env import BatchedEnv  # imports torch; fails loudly without a GPU
            backend = BatchedEnv(self.cfg, device)
        self.backend = backend
        # te_set_persistent_obs: the LIDAR observation is updated in place instead of being streamed whole every step.  With numpy output
        # the device buffers never leave this object (the caller gets host copies), so the promise "nobody else writes them" holds by
        # construction and the mode is on unless asked otherwise; with torch output the caller holds the device tensors: opt-in.
        if persistent_obs is None:
            persistent_obs = output == "numpy"
        if persistent_obs and hasattr(backend, "set_persistent_obs"):
            backend.set_persistent_obs(True)
        self.num_envs = int(num_envs)
        self.stacked = bool(self.cfg.stacked_obs)  # level5: stacked_spheres + validity_mask instead of lidar
        self.observation_space = spaces.stacked_observation_space() if self.stacked else spaces.observation_space((int(self.cfg.lidar_channels), K.LIDAR_NTHETA, K.LIDAR_NPHI))
        self.action_space = spaces.action_space()
        self.render_mode = None
        self.reset_infos: List[Dict[str, Any]] = [{} for _ in range(self.num_envs)]
        self._actions = None
        self._seeds: List[Optional[int]] = [None] * self.num_envs
        self._options: List[Dict[str, Any]] = [{} for _ in range(self.num_envs)]
        self.metadata = {"render_modes": []}
        # exp05: the ally's driver (exp05_vFinal_environment.py:103-104); set with update_model / env_method("update_model", m)
        self.external_ally = int(self.cfg.ally_policy) == K.ALLY_EXTERNAL
        self.lw_driver = None

    # ------------------------------------------------------------------ helpers
    def _to_host(self, tensors):
        """Device tensors -> numpy arrays with ONE drain of the stream: each lands in a fresh pinned block of torch's caching host allocator
        (the arrays are the caller's to keep; blocks of dropped arrays are reused), copied asynchronously.  `t.cpu()` per tensor went through
        pageable memory: 33 ms per step at 65 536 envs instead of 5 (tools/vecenv_bench.py)."""
        import torch

        outs = []
        on_gpu = False
        for t in tensors:
            t = t.detach()
            if t.device.type == "cpu":          # stub backends of the CPU tests
                outs.append(t)
                continue
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t, non_blocking=True)
            outs.append(h)
            on_gpu = True
        if on_gpu:
            torch.cuda.current_stream(self.backend.device).synchronize()
        return [h.numpy() for h in outs]

    def _snapshot_async(self, tensors):
        """Enqueue non-blocking copies of small device tensors into fresh pinned blocks and record an event; the returned callable waits for
        that event (not for the stream) and hands out the numpy arrays."""
        import torch

        if all(t.device.type == "cpu" for t in tensors):          # stub backends of the CPU tests
            held = [t.detach().clone() for t in tensors]
            return lambda: [h.numpy() for h in held]
        held = []
        for t in tensors:
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t.detach(), non_blocking=True)
            held.append(h)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.backend.device))

        def wait():
            ev.synchronize()
            return [h.numpy() for h in held]
        return wait

    def _obs_keys(self):
        return ("stacked_spheres", "validity_mask", "inertial_data", "last_action") if self.stacked else ("lidar", "inertial_data", "last_action")

    def _obs(self, *t):
        if self.stacked:
            t = (t[0], t[1].bool(), t[2], t[3])
        return dict(zip(self._obs_keys(), t if self.output == "torch" else self._to_host(t)))

    def _as_device_actions(self, actions):
        import torch

        b = self.backend
        if isinstance(actions, torch.Tensor):
            return actions.to(device=b.device, dtype=torch.float32).reshape(self.num_envs, 4)
        a = np.asarray(actions, dtype=np.float32).reshape(self.num_envs, 4)
        return torch.from_numpy(np.ascontiguousarray(a)).to(b.device)

    # ------------------------------------------------------------------ VecEnv API
    def reset(self):
        obs = self.backend.reset()
        if self.stacked:
            obs = self.backend.observe_stacked()
        self.reset_infos = [{} for _ in range(self.num_envs)]
        self._seeds = [None] * self.num_envs
        self._options = [{} for _ in range(self.num_envs)]
        return self._obs(*obs)

    def step_async(self, actions) -> None:
        self._actions = self._as_device_actions(actions)

    def step_wait(self):
        if self._actions is None:
            raise RuntimeError("step_wait() without step_async()")
        b = self.backend
        if self.external_ally:
            self._drive_ally()
        *obs_t, reward, done, info = (b.step_stacked if self.stacked else b.step)(self._actions, terminal=True)
        self._actions = None
        if self.stacked:
            obs_t = [obs_t[0], obs_t[1].bool(), obs_t[2], obs_t[3]]
        tbuf = ({"stacked_spheres": b.t_stacked, "validity_mask": b.t_mask.bool()} if self.stacked else {"lidar": b.t_lidar})
        tbuf.update({"inertial_data": b.t_inertial, "last_action": b.t_last_action})
        if self.output == "torch":
            # The observations stay on the device.  done / info (17 bytes per env) are the backend's persistent buffers, which the NEXT step
            # overwrites: their values of THIS step are copied now, asynchronously, into fresh pinned memory behind an event; an info indexed
            # later (a logger that drains afterwards, an asynchronous collector) waits for that event only and reads this step's values.
            host = {}
            fetch = self._snapshot_async([info, done])

            def both():
                if not host:
                    host["info"], d = fetch()
                    host["done"] = d.astype(bool)
                return host
            gen = getattr(b, "generation", None)   # (backends without a step counter: no lifetime check)
            infos = LazyInfos(lambda: both()["info"], lambda: both()["done"], tbuf,
                              still_valid=None if gen is None else (lambda: b.generation == gen))
            if self.infos_mode == "dicts":
                infos = infos.materialise()
            return dict(zip(self._obs_keys(), obs_t)), reward, done.bool(), infos
        *obs_np, rew, done_u8, info_np = self._to_host([*obs_t, reward, done, info])
        done_np = done_u8.astype(bool)
        terminal = None
        if done_np.any():  # copy only the rows that are valid
            import torch

            idx = np.flatnonzero(done_np)
            ti = torch.from_numpy(idx).to(b.device)
            rows = self._to_host([v[ti] for v in tbuf.values()])
            terminal = {k: _ScatterRows(idx, v, self.num_envs) for k, v in zip(tbuf.keys(), rows)}
        infos = LazyInfos(info_np, done_np, terminal)
        if self.infos_mode == "dicts":
            infos = infos.materialise()
        return dict(zip(self._obs_keys(), obs_np)), rew, done_np, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    # ------------------------------------------------------------------ exp05
    def update_model(self, model) -> None:
        """The ally's policy: SB3's `predict(observation, deterministic=True) -> (actions [N,4], state)` on the batched
        observation dict of the allies.  A driver with a true `accepts_torch` attribute gets device tensors (no host
        copy: dronechase_amd.ppo.PolicyDriver); anything else gets numpy arrays, like an SB3 model."""
        if not self.external_ally:
            raise AttributeError("update_model: only exp05 (cfg.ally_policy == ALLY_EXTERNAL) has an ally driver")
        self.lw_driver = model

    def _drive_ally(self) -> None:
        # Exp05_vFinal_Task.drive_lw_rl_agent (exp05_vFinal_task.py:252-260), for all envs at once
        if self.lw_driver is None:
            raise AttributeError("exp05: step before update_model(model): the ally has no driver")
        import torch

        lidar, inertial, last_action, _active = self.backend.observe_ally()
        obs = {"lidar": lidar, "inertial_data": inertial, "last_action": last_action}
        if not getattr(self.lw_driver, "accepts_torch", False):
            obs = {k: v.detach().cpu().numpy() for k, v in obs.items()}
        actions, _ = self.lw_driver.predict(obs, deterministic=True)
        self.backend.set_ally_actions(self._as_device_actions(actions).contiguous())

    def close(self) -> None:
        if self.backend is not None:
            self.backend.close()

    def seed(self, seed: Optional[int] = None):
        """The reference ignores seeds (exp03_vFinal_environment.py:128 `reset(self, seed=0)`); here the seed
        is fixed at construction because it keys the device RNG.  Returns the per-env list SB3 expects."""
        return [None] * self.num_envs

    def _indices(self, indices) -> Iterable[int]:
        if indices is None:
            return range(self.num_envs)
        if isinstance(indices, int):
            return [indices]
        return indices

    def get_attr(self, attr_name: str, indices=None) -> List[Any]:
        n = len(list(self._indices(indices)))
        if attr_name == "render_mode":
            return [None] * n
        if hasattr(self, attr_name):
            return [getattr(self, attr_name)] * n
        raise AttributeError(attr_name)

    def set_attr(self, attr_name: str, value: Any, indices=None) -> None:
        raise AttributeError(f"per-env attribute {attr_name!r} cannot be set on a batched environment")

    def env_method(self, method_name: str, *method_args, indices=None, **method_kwargs) -> List[Any]:
        if method_name == "update_model" and self.external_ally:  # one driver for the whole batch
            self.update_model(*method_args, **method_kwargs)
            return [None] * len(list(self._indices(indices)))
        raise AttributeError(f"per-env method {method_name!r} is not available on a batched environment")

    def env_is_wrapped(self, wrapper_class, indices=None) -> List[bool]:
        return [False] * len(list(self._indices(indices)))

    def get_images(self):
        return [None] * self.num_envs

    def render(self, mode: Optional[str] = None):
        return None

    @property
    def unwrapped(self):
        return self

    def __len__(self):
        return self.num_envs


class _ScatterRows:
    """Read-only view mapping env index -> row of a compact [n_done, ...] array."""

    def __init__(self, idx: np.ndarray, rows: np.ndarray, n: int):
        self._pos = {int(e): k for k, e in enumerate(idx)}
        self._rows = rows

    def __getitem__(self, env: int):
        return self._rows[self._pos[int(env)]]
