"""Mirror of ReinforcementLearningPipeline.create_vectorized_environment
(src/core/rl_framework/utils/pipeline.py:31-61): same arguments and kwarg white-list, but returns ONE
batched GPU VecEnv instead of VecMonitor(SubprocVecEnv([env]*n))."""
from __future__ import annotations

import os
from typing import Optional

from .envs import ENV_TASKS
from .vec_env import ThreatEngageVecEnv


class ReinforcementLearningPipeline:
    @staticmethod
    def create_vectorized_environment(environment, env_kwargs: Optional[dict] = None, n_envs: int = os.cpu_count() or 1,
                                      GUI: bool = False, env_args=None, device: str = "cuda:0", monitor: bool = True,
                                      **vec_kwargs):
        env_kwargs = dict(env_kwargs or {})
        n_envs = n_envs if not GUI else 1
        env_kwargs["GUI"] = GUI
        env_args = ["dome_radius", "rl_frequency", "GUI"] if env_args is None else env_args
        valid = {k: v for k, v in env_kwargs.items() if k in env_args}
        task = ENV_TASKS.get(environment, environment if isinstance(environment, str) else None)
        if task is None:
            raise ValueError(f"{environment!r} is not one of the environments of this hot path: {sorted(c.__name__ for c in ENV_TASKS)}")
        venv = ThreatEngageVecEnv(task=task, num_envs=n_envs, device=device, **valid, **vec_kwargs)
        if monitor:
            try:  # VecMonitor wraps any VecEnv (pipeline.py:61); optional because SB3 may be absent
                from stable_baselines3.common.vec_env import VecMonitor  # type: ignore
                return VecMonitor(venv)
            except Exception:
                pass
        return venv
