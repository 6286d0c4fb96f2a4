"""BatchedEnv: thin PyTorch-ROCm front-end of the C ABI.  Torch is plumbing only (device memory and
streams); every computation runs in the HIP kernels of libthreatengage.so."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from . import config as K


class BatchedEnv:
    """N environments of one task resident on one MI355X.

    Observation tensors (views of internal device buffers, overwritten by the next call):
      lidar [N,3,13,26] f32, inertial [N,15] f32, last_action [N,4] f32
    """

    def __init__(self, cfg: K.Config, device: str | torch.device = "cuda:0"):
        self.L = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.TEError("BatchedEnv needs a HIP device ('cuda:N'); there is no CPU path")
        if not torch.cuda.is_available():
            raise _lib.TEError("no GPU visible to PyTorch-ROCm; dronechase_amd has no CPU fallback")
        self.cfg = cfg.copy()
        self.N, self.D = int(cfg.n_envs), cfg.n_drones
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._h = C.c_void_p()
        _lib.check(self.L.te_create(C.byref(self.cfg), index, C.byref(self._h)), "te_create")
        N, dev = self.N, self.device
        f32 = dict(dtype=torch.float32, device=dev)
        self.lidar = torch.empty((N, int(cfg.lidar_channels), K.LIDAR_NTHETA, K.LIDAR_NPHI), **f32)
        self.inertial = torch.empty((N, K.OBS_INERTIAL_WORDS), **f32)
        self.last_action = torch.empty((N, 4), **f32)
        self.t_lidar = torch.zeros_like(self.lidar)
        self.t_inertial = torch.zeros_like(self.inertial)
        self.t_last_action = torch.zeros_like(self.last_action)
        self.reward = torch.empty((N,), **f32)
        self.done = torch.empty((N,), dtype=torch.uint8, device=dev)
        self.info = torch.empty((N, K.INFO_WORDS), dtype=torch.int32, device=dev)
        self.generation = 0   # number of env.step calls so far: the t_* (terminal) buffers hold the rows of the LAST step only
        self.stacked_mode = bool(cfg.stacked_obs)
        if self.stacked_mode:  # level5: FusedLIDAR stacked observation instead of the own sphere
            shape = (N, K.STACK_SPHERES, K.LIDAR_CHANNELS, K.LIDAR_NTHETA, K.LIDAR_NPHI)
            self.stacked = torch.empty(shape, **f32)
            self.mask = torch.empty((N, K.STACK_SPHERES), dtype=torch.uint8, device=dev)
            self.t_stacked = torch.zeros(shape, **f32)
            self.t_mask = torch.zeros((N, K.STACK_SPHERES), dtype=torch.uint8, device=dev)

    # ------------------------------------------------------------------ plumbing
    def _stream(self) -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _p(t: Optional[torch.Tensor]) -> C.c_void_p:
        return C.c_void_p(0 if t is None else t.data_ptr())

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            torch.cuda.synchronize(self.device)
            self.L.te_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check_actions(self, actions: torch.Tensor) -> torch.Tensor:
        if actions.device != self.device or actions.dtype != torch.float32 or tuple(actions.shape) != (self.N, 4):
            raise ValueError(f"actions must be float32 [{self.N}, 4] on {self.device}")
        return actions.contiguous()

    # ------------------------------------------------------------------ API
    def observe(self, out=None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        lidar, inertial, last_action = self._obs_out(out)
        _lib.check(self.L.te_observe(self._h, self._p(lidar), self._p(inertial), self._p(last_action), self._stream()), "te_observe")
        return lidar, inertial, last_action

    def reset(self, mask: Optional[torch.Tensor] = None):
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            if mask.numel() != self.N:
                raise ValueError("mask must have n_envs entries")
        _lib.check(self.L.te_reset(self._h, self._p(mask), self._stream()), "te_reset")
        if self.D > 32:   # te_observe serves 32 drones: Level5DumbMultiObs (37, all scripted) observes through step_students, Level5FusionEnvironment (36) through observe_stacked
            return None if (self.cfg.agent_scripted or self.cfg.evaluation) else self.observe_stacked()
        return self.observe()

    def _obs_out(self, out):
        """(lidar, inertial, last_action) destination tensors of this step: the internal buffers, or the caller's (e.g. slot t
        of a rollout buffer in HBM: the observation is written where it will be kept, no copy)."""
        if out is None:
            return self.lidar, self.inertial, self.last_action
        for o, ref, align in zip(out, (self.lidar, self.inertial, self.last_action), (16, 4, 16)):
            if o.shape != ref.shape or o.dtype != ref.dtype or o.device != ref.device or not o.is_contiguous() or o.data_ptr() % align:
                raise ValueError("out = (lidar [N,3,13,26], inertial [N,15], last_action [N,4]): contiguous float32 on the env's device, "
                                 "lidar and last_action 16-byte aligned")
        return tuple(out)

    def step(self, actions: torch.Tensor, terminal: bool = True, out=None):
        a = self._check_actions(actions)
        t = (self.t_lidar, self.t_inertial, self.t_last_action) if terminal else (None, None, None)
        lidar, inertial, last_action = self._obs_out(out)
        _lib.check(self.L.te_step(self._h, self._p(a), self._p(lidar), self._p(inertial),
                                  self._p(last_action), self._p(self.reward), self._p(self.done), self._p(self.info),
                                  self._p(t[0]), self._p(t[1]), self._p(t[2]), self._stream()), "te_step")
        self.generation += 1
        return lidar, inertial, last_action, self.reward, self.done, self.info

    # level5 ---------------------------------------------------------------------------------
    def observe_stacked(self):
        _lib.check(self.L.te_observe_stacked(self._h, self._p(self.stacked), self._p(self.mask), self._p(self.inertial),
                                             self._p(self.last_action), self._stream()), "te_observe_stacked")
        return self.stacked, self.mask, self.inertial, self.last_action

    def step_stacked(self, actions: torch.Tensor, terminal: bool = True):
        a = self._check_actions(actions)
        t = (self.t_stacked, self.t_mask, self.t_inertial, self.t_last_action) if terminal else (None,) * 4
        _lib.check(self.L.te_step_stacked(self._h, self._p(a), self._p(self.stacked), self._p(self.mask), self._p(self.inertial),
                                          self._p(self.last_action), self._p(self.reward), self._p(self.done), self._p(self.info),
                                          self._p(t[0]), self._p(t[1]), self._p(t[2]), self._p(t[3]), self._stream()), "te_step_stacked")
        self.generation += 1
        return self.stacked, self.mask, self.inertial, self.last_action, self.reward, self.done, self.info

    def set_persistent_obs(self, on: bool = True) -> None:
        """Promise that nobody but the library writes the LIDAR observation buffers.  While the same buffer keeps being passed (this
        object's own `lidar` / `stacked` / `_students` tensors are, unless step(out=...) names another one), a step rewrites only the
        cells that change instead of streaming the whole background first (te_set_persistent_obs).  The library recognises "the same buffer" by
        its ADDRESS: a tensor passed through step(out=...) that is freed and whose memory the caching allocator hands to a new tensor of the
        same shape would be taken for the old one.  Keep the tensor alive for as long as it is passed, or switch the mode off and on again
        (which forgets the buffer) when the out tensor changes."""
        _lib.check(self.L.te_set_persistent_obs(self._h, 1 if on else 0), "te_set_persistent_obs")

    def step_students(self):
        """Level5DumbMultiObs (every wingman scripted): one env.step returning the student observation of EVERY pursuer and the teacher's
        action for it: (stacked [N,P,6,3,13,26], mask [N,P,6], inertial [N,P,15], last_action [N,P,4], active [N,P], reward, done, info)."""
        if not hasattr(self, "_students"):
            N, P, dev = self.N, int(self.cfg.n_pursuers), self.device
            f32 = dict(dtype=torch.float32, device=dev)
            self._students = (torch.empty((N, P, K.STACK_SPHERES, K.LIDAR_CHANNELS, K.LIDAR_NTHETA, K.LIDAR_NPHI), **f32),
                              torch.empty((N, P, K.STACK_SPHERES), dtype=torch.uint8, device=dev), torch.empty((N, P, K.OBS_INERTIAL_WORDS), **f32),
                              torch.empty((N, P, 4), **f32), torch.empty((N, P), dtype=torch.uint8, device=dev))
        st = self._students
        _lib.check(self.L.te_step_students(self._h, self._p(st[0]), self._p(st[1]), self._p(st[2]), self._p(st[3]), self._p(st[4]),
                                           self._p(self.reward), self._p(self.done), self._p(self.info), self._stream()), "te_step_students")
        return (*st, self.reward, self.done, self.info)

    # caller-driven wingmen (exp05's ally, "nn" drivers of the evaluation task) -----------------
    def observe_wingman(self, wingman: int):
        """Observation of a caller-driven pursuer on the current state (compute_lw_observation, exp05_vFinal_task.py:265-292,
        evaluation_task.py:281-310): (lidar [N,3,13,26], inertial [N,15], last_action [N,4], active [N] u8)."""
        if not hasattr(self, "_wingman_buf"):
            self._wingman_buf = {}
        if wingman not in self._wingman_buf:
            self._wingman_buf[wingman] = (torch.empty_like(self.lidar), torch.empty_like(self.inertial), torch.empty_like(self.last_action),
                                          torch.empty((self.N,), dtype=torch.uint8, device=self.device))
        lidar, inertial, last_action, active = self._wingman_buf[wingman]
        _lib.check(self.L.te_observe_wingman(self._h, int(wingman), self._p(lidar), self._p(inertial), self._p(last_action), self._p(active),
                                             self._stream()), "te_observe_wingman")
        return lidar, inertial, last_action, active

    def set_wingman_actions(self, wingman: int, actions: torch.Tensor) -> None:
        """`pursuer.drive(action)` for every env whose pursuer `wingman` is armed."""
        a = self._check_actions(actions)
        _lib.check(self.L.te_set_wingman_actions(self._h, int(wingman), self._p(a), self._stream()), "te_set_wingman_actions")

    def observe_ally(self):
        """exp05: the ally = pursuer 1 (Exp05_vFinal_Task.compute_lw_observation)."""
        if int(self.cfg.ally_policy) != K.ALLY_EXTERNAL:
            _lib.check(self.L.te_observe_ally(self._h, None, None, None, None, self._stream()), "te_observe_ally")  # raises with the library's message
        self.ally_lidar, self.ally_inertial, self.ally_last_action, self.ally_active = self.observe_wingman(1)
        return self.ally_lidar, self.ally_inertial, self.ally_last_action, self.ally_active

    def set_ally_actions(self, actions: torch.Tensor) -> None:
        """`pursuer.drive(action)` for every env whose ally is armed (exp05_vFinal_task.py:255-260)."""
        a = self._check_actions(actions)
        _lib.check(self.L.te_set_ally_actions(self._h, self._p(a), self._stream()), "te_set_ally_actions")

    def wingman_info(self) -> torch.Tensor:
        """[N,P,5] i32 rows (lw_kills, lw_alive, lw_munitions, current_wave, step) of every pursuer
        (Evaluation_Task.compute_info, evaluation_task.py:553-574); cfg.evaluation only."""
        if not hasattr(self, "_wingman_info"):
            self._wingman_info = torch.empty((self.N, int(self.cfg.n_pursuers), 5), dtype=torch.int32, device=self.device)
        _lib.check(self.L.te_wingman_info(self._h, self._p(self._wingman_info), self._stream()), "te_wingman_info")
        return self._wingman_info

    def random_actions(self, seed: int, step_index: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if out is None:
            out = torch.empty((self.N, 4), dtype=torch.float32, device=self.device)
        _lib.check(self.L.te_random_actions(self._h, self._p(out), seed, step_index, self._stream()), "te_random_actions")
        return out

    def state_words(self) -> int:
        n = C.c_size_t()
        _lib.check(self.L.te_state_words(self._h, C.byref(n)), "te_state_words")
        return int(n.value)

    def get_state(self) -> torch.Tensor:
        """State blob (include/threatengage.h TE_D_*/TE_E_*) as an int32 tensor of raw 4-byte words."""
        w = torch.empty((self.state_words(),), dtype=torch.int32, device=self.device)
        _lib.check(self.L.te_get_state(self._h, self._p(w), w.numel(), self._stream()), "te_get_state")
        return w

    def set_state(self, words: torch.Tensor) -> None:
        w = words.to(device=self.device).contiguous()
        if w.dtype not in (torch.int32, torch.uint32) or w.numel() != self.state_words():
            raise ValueError("state must be te_state_words() 4-byte words")
        _lib.check(self.L.te_set_state(self._h, self._p(w), w.numel(), self._stream()), "te_set_state")
        torch.cuda.current_stream(self.device).synchronize()  # `w` may be a temporary

    def profile_begin(self, max_steps: int) -> None:
        _lib.check(self.L.te_profile_begin(self._h, max_steps), "te_profile_begin")

    def profile_end(self) -> Tuple[float, float, int]:
        a, b, n = C.c_float(), C.c_float(), C.c_int32()
        _lib.check(self.L.te_profile_end(self._h, C.byref(a), C.byref(b), C.byref(n)), "te_profile_end")
        return float(a.value), float(b.value), int(n.value)
