"""Single-environment classes with the reference's names and constructor signature
`(dome_radius, rl_frequency, GUI)`: gymnasium-style reset(seed) -> (obs, info), step(a) -> 5-tuple.

They are N=1 views of the batched GPU environment (no separate implementation); for throughput use
ThreatEngageVecEnv / create_vectorized_environment."""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import config as K
from . import spaces
from .vec_env import INFO_KEYS, make_config


class _SingleEnv:
    TASK = "stage03"
    DEFAULT_DOME = 20.0

    def __init__(self, dome_radius: Optional[float] = None, rl_frequency: int = 15, GUI: bool = False,
                 device: str = "cuda:0", seed: int = 0, **overrides):
        if GUI:
            raise ValueError("GUI rendering is out of scope (PyBullet debug window)")
        from .batched_env import BatchedEnv

        self.dome_radius = self.DEFAULT_DOME if dome_radius is None else float(dome_radius)
        self.rl_frequency = rl_frequency
        self.cfg = make_config(self.TASK, 1, self.dome_radius, rl_frequency, seed, auto_reset=0, **overrides)
        self._b = BatchedEnv(self.cfg, device)
        self.action_space = spaces.action_space()
        self.stacked = bool(self.cfg.stacked_obs)
        self.observation_space = spaces.stacked_observation_space() if self.stacked else spaces.observation_space((int(self.cfg.lidar_channels), K.LIDAR_NTHETA, K.LIDAR_NPHI))

    def _obs(self, *t):
        if self.stacked:
            stacked, mask, inertial, last_action = t
            return {"stacked_spheres": stacked[0].cpu().numpy(), "validity_mask": mask[0].cpu().numpy().astype(bool),
                    "inertial_data": inertial[0].cpu().numpy(), "last_action": last_action[0].cpu().numpy()}
        lidar, inertial, last_action = t
        return {"lidar": lidar[0].cpu().numpy(), "inertial_data": inertial[0].cpu().numpy(),
                "last_action": last_action[0].cpu().numpy()}

    def reset(self, seed=0, options=None):
        """`seed` is accepted and ignored, as in the reference (exp03_vFinal_environment.py:128-146)."""
        self._b.reset()
        return self._obs(*(self._b.observe_stacked() if self.stacked else self._b.observe())), {}

    def step(self, rl_action: np.ndarray):
        import torch

        a = torch.as_tensor(np.asarray(rl_action, np.float32).reshape(1, 4), device=self._b.device)
        *obs, reward, done, info = (self._b.step_stacked if self.stacked else self._b.step)(a, terminal=False)
        inf = dict(zip(INFO_KEYS, (int(v) for v in info[0].cpu().numpy())))
        return self._obs(*obs), float(reward[0].item()), bool(done[0].item()), False, inf

    def close(self):
        self._b.close()


class Exp02vFinalEnvironment(_SingleEnv):  # level4/exp02_vFinal_environment.py:30
    TASK = "exp02"


class Exp03vFinalEnvironment(_SingleEnv):  # level4/exp03_vFinal_environment.py:28
    TASK = "exp03"


class Exp04vFinalEnvironment(_SingleEnv):  # level4/exp04_vFinal_environment.py:22
    TASK = "exp04"


class Exp05vFinalEnvironment(_SingleEnv):  # level4/exp05_vFinal_environment.py
    """exp03 with the ally flown by a second policy.  `update_model(model)` (exp05_vFinal_environment.py:103-104) takes
    anything with SB3's `predict(observation, deterministic=True) -> (action, state)`; as in the reference
    (Exp05_vFinal_Task.drive_lw_rl_agent, exp05_vFinal_task.py:252-260) it is asked once per step, before the physics,
    with the ally's own observation, and stepping without a model is an error (the reference raises AttributeError)."""
    TASK = "exp05"

    def update_model(self, model) -> None:
        self.lw_driver = model

    def step(self, rl_action: np.ndarray):
        import torch

        if not hasattr(self, "lw_driver"):
            raise AttributeError("Exp05vFinalEnvironment.step before update_model(model): the ally has no driver")
        lidar, inertial, last_action, active = self._b.observe_ally()
        if bool(active[0].item()):
            obs = {"lidar": lidar[0].cpu().numpy(), "inertial_data": inertial[0].cpu().numpy(), "last_action": last_action[0].cpu().numpy()}
            action, _ = self.lw_driver.predict(obs, deterministic=True)
            self._b.set_ally_actions(torch.as_tensor(np.asarray(action, np.float32).reshape(1, 4), device=self._b.device))
        return super().step(rl_action)


class EvaluationEnvironment(_SingleEnv):  # level4/evaluation_environment.py (behaviour-tree drivers)
    """`EvaluationEnvironment(configuration, GUI, rl_frequency)`: every wingman is flown by the task
    (evaluation_environment.py:170-187); `configuration["drivers"]` is the reference's list of
    `{"type": "bt", "name": ...}` (evaluation_exp01_1bt_app_ready.py:64-68).  A driver of type "nn" loads a
    stable-baselines3 checkpoint from `"path"` in the reference (`PPO.load`, evaluation_task.py:658-661); SB3 is not
    installable here, so it takes the loaded object instead: `{"type": "nn", "name": ..., "model": obj}` with SB3's
    `predict(observation, deterministic=True)`, asked once per step with that wingman's own observation, as
    Evaluation_Task.drive_lw does (:257-268).  `step` returns reward 0 and the per-wingman info rows
    `{name: {"lw_kills", "lw_alive", "lw_munitions", "current_wave", "step"}}` of the armed wingmen (:553-574)."""
    TASK = "evaluation"
    INFO_KEYS = ("lw_kills", "lw_alive", "lw_munitions", "current_wave", "step")

    def __init__(self, configuration: Optional[dict] = None, GUI: bool = False, rl_frequency: int = 15, dome_radius: Optional[float] = None,
                 **overrides):
        configuration = dict(configuration or {})
        drivers = configuration.get("drivers", [{"type": "bt", "name": "bt_1"}])
        for d in drivers:
            if d.get("type") == "nn" and not hasattr(d.get("model"), "predict"):
                raise ValueError("EvaluationEnvironment: a driver of type 'nn' needs 'model': an object with SB3's predict() "
                                 "(the reference's 'path' is a stable-baselines3 checkpoint: load it where SB3 is installed)")
            if d.get("type") not in ("bt", "nn"):
                raise ValueError("EvaluationEnvironment: drivers are behaviour-tree ({'type': 'bt'}) or policy ({'type': 'nn', 'model': ...}) wingmen")
        self.driver_names = [d.get("name", f"{d.get('type')}_{i + 1}") for i, d in enumerate(drivers)]
        self.models = {p: d["model"] for p, d in enumerate(drivers) if d.get("type") == "nn"}
        overrides["evaluation"] = 1 | (sum(1 << p for p in self.models) << 8)
        from . import _lib
        P, mun = len(drivers), int(configuration.get("munition_per_defender", 20))
        rounds = int(_lib.load().te_calculate_rounds(P, mun))  # evaluation_task.py:96-100
        limited = bool(configuration.get("TIME_IS_LIMITED", False))
        overrides.update(n_pursuers=P, munition=mun, n_rounds=rounds, n_invaders=rounds,
                         born_radius=float(configuration.get("ENEMY_BORN_RADIUS", 6)),
                         step_increment=int(configuration.get("STEP_INCREMENT", 100)),
                         max_step=int(configuration.get("MAX_STEP", 300)) if limited else 0)
        super().__init__(dome_radius, rl_frequency, GUI, **overrides)

    def step(self, actions_not_used=None):
        import torch

        for p, model in self.models.items():   # Evaluation_Task.drive_lw: observation -> predict -> drive, armed wingmen only
            lidar, inertial, last_action, active = self._b.observe_wingman(p)
            if bool(active[0].item()):
                o = {"lidar": lidar[0].cpu().numpy(), "inertial_data": inertial[0].cpu().numpy(), "last_action": last_action[0].cpu().numpy()}
                action, _ = model.predict(o, deterministic=True)
                self._b.set_wingman_actions(p, torch.as_tensor(np.asarray(action, np.float32).reshape(1, 4), device=self._b.device))
        obs, reward, done, truncated, _ = super().step(np.zeros(4, np.float32))
        rows = self._b.wingman_info()[0].cpu().numpy()
        info = {name: dict(zip(self.INFO_KEYS, (int(rows[p, 0]), bool(rows[p, 1]), *(int(x) for x in rows[p, 2:]))))
                for p, name in enumerate(self.driver_names) if rows[p, 1]}
        return obs, reward, done, truncated, info


class PyflytL2EnviromentModifiedV2(_SingleEnv):  # level2/pyflyt_level2_environment_modified_v2.py:15 (sic)
    TASK = "stage01"
    DEFAULT_DOME = 10.0


class PyflytL3EnviromentV2(_SingleEnv):  # level3/pyflyt_level3_environment_v2.py:20 (sic)
    TASK = "stage02"
    DEFAULT_DOME = 8.0


class Level5Environment(_SingleEnv):  # threatsense/level5/level5_envrionment.py:32 (student observation)
    TASK = "level5"


class Level5DumbMultiObs(_SingleEnv):  # threatsense/level5/level5_dumb_multiobs.py:10
    """The imitation-data collector's environment (apps/threatsense_runner/collect_and_save.py): seven wingmen, all flown by the behaviour
    tree, 5 -> 30 invaders.  As in the reference the observation is a dummy `zeros(1)`, `step` ignores its action, and everything of
    interest travels in `info`: `student_observations` = the stacked observation (stacked_spheres [6,3,13,26], validity_mask [6],
    inertial_data [15], last_action [4]) of every ARMED wingman, `teacher_actions` = their behaviour-tree commands (:116-150).  The batched
    form is `BatchedEnv(default_config("level5_dumb", ...)).step_students()` -> [N,7,...] tensors on the device + the `active` mask."""
    TASK = "level5_dumb"

    def __init__(self, GUI: bool = False, rl_frequency: int = 15, dome_radius: Optional[float] = None, **overrides):
        super().__init__(dome_radius, rl_frequency, GUI, **overrides)
        from . import spaces as S
        self.observation_space = S.Box(0, 1, shape=(1,), dtype=np.float32)      # level5_dumb_multiobs.py:27-33

    def reset(self, seed=0, options=None):
        self._b.reset()
        return np.zeros(1, np.float32), {"student_observations": [], "teacher_actions": []}

    def step(self, action=None):
        stacked, mask, inertial, last_action, active, reward, done, _info = (t[0].cpu().numpy() if t.dim() > 1 else t.cpu().numpy() for t in self._b.step_students())
        rows = [p for p in range(stacked.shape[0]) if active[p]]
        students = [{"stacked_spheres": stacked[p], "validity_mask": mask[p].astype(bool), "inertial_data": inertial[p], "last_action": last_action[p]} for p in rows]
        info = {"student_observations": students, "teacher_actions": [last_action[p] for p in rows]}
        return np.zeros(1, np.float32), float(reward[0]), bool(done[0]), False, info


class Level5C1FusionEnvironment(_SingleEnv):  # threatsense/level5/level5_c1_fusion_environment.py:7
    """The agent and one scripted wingman against 4 -> 10 invaders, the student observation of Level5Environment, Level5C1FusionTask's
    minimal reward; `info` is empty (level5_c1_fusion_environment.py:105-106)."""
    TASK = "level5_c1"

    def __init__(self, GUI: bool = False, rl_frequency: int = 15, dome_radius: Optional[float] = None, **overrides):
        super().__init__(dome_radius, rl_frequency, GUI, **overrides)

    def step(self, rl_action: np.ndarray):
        obs, reward, done, truncated, _ = super().step(rl_action)
        return obs, reward, done, truncated, {}


class Level5FusionEnvironment(_SingleEnv):  # threatsense/level5/level5_fusion_environment.py:5
    """Level5Environment with Level5FusionTask: the agent + five scripted wingmen, 5 -> 30 invaders in steps of five (36 drones)."""
    TASK = "level5_fusion"

    def __init__(self, GUI: bool = False, rl_frequency: int = 15, dome_radius: Optional[float] = None, **overrides):
        super().__init__(dome_radius, rl_frequency, GUI, **overrides)


class Level52BTEvaluationEnvironment(_SingleEnv):  # threatsense/level5/level5_eval_2bt_environment.py:12
    """Two behaviour-tree wingmen against the 30-slot invader table (Level52BTEvaluationTask).  As in the reference `reset` and `step`
    return an EMPTY observation (`{}`: level5_eval_2bt_environment.py:53-56,75), reward 0.0, and the info of
    Level52BTEvaluationTask.compute_info (level5_2bt_evaluation_task.py:470-477): `kills_per_drone` = {drone id: {"name", "type": "BT",
    "kills"}} (ids 0 and 1 here: the reference's are PyBullet body ids), `deads`, `current_wave`."""
    TASK = "level5_2bt"
    NAMES = ("Ally1", "Ally2")   # spawn_pursuer_squad (level5_2bt_evaluation_task.py:541-547): every pursuer is an "Ally{i}"

    def __init__(self, GUI: bool = False, rl_frequency: int = 15, dome_radius: Optional[float] = None, **overrides):
        super().__init__(dome_radius, rl_frequency, GUI, **overrides)
        from . import spaces as S
        self.observation_space = S.Box(0, 1, shape=(1,), dtype=np.float32)      # level5_eval_2bt_environment.py:27-29
        self._zero = None

    def _info(self):
        rows = self._b.wingman_info()[0].cpu().numpy()     # (kills, alive, munition, wave, step) per wingman
        deads = int(self._b.get_state()[self._b.cfg.n_drones * K.DRONE_WORDS + K.E["DEADS"]].item())
        return {"kills_per_drone": {p: {"name": self.NAMES[p], "type": "BT", "kills": int(rows[p, 0])} for p in range(rows.shape[0])},
                "deads": deads, "current_wave": int(rows[0, 3])}

    def reset(self, seed=0, options=None):
        self._b.reset()
        return {}, self._info()

    def step(self, action=None):
        import torch

        if self._zero is None:   # the action is not used (level5_eval_2bt_environment.py:41-58): both wingmen obey the tree
            self._zero = torch.zeros((1, 4), dtype=torch.float32, device=self._b.device)
        done = self._b.step(self._zero, terminal=False)[4]
        return {}, 0.0, bool(done[0].item()), False, self._info()


ENV_TASKS = {cls: cls.TASK for cls in (Level5DumbMultiObs, Level52BTEvaluationEnvironment, Level5C1FusionEnvironment, Level5FusionEnvironment, Exp02vFinalEnvironment, Exp03vFinalEnvironment, Exp04vFinalEnvironment, Exp05vFinalEnvironment, EvaluationEnvironment,
                                       PyflytL2EnviromentModifiedV2, PyflytL3EnviromentV2, Level5Environment)}
