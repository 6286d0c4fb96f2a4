"""Observation / action spaces.  gymnasium is used when importable; otherwise this 40-line shim with the
same attribute surface (SB3's preprocessing only needs .shape, .dtype, .low, .high, .spaces, .sample)."""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - depends on the box
    from gymnasium.spaces import Box, Dict  # type: ignore
    HAVE_GYMNASIUM = True
except Exception:  # gymnasium is absent in the build container and on the GPU box
    HAVE_GYMNASIUM = False

    class Box:  # type: ignore
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.dtype = np.dtype(dtype)
            if shape is None:
                shape = np.shape(low)
            self.shape = tuple(shape)
            self.low = np.broadcast_to(np.asarray(low, self.dtype), self.shape).copy()
            self.high = np.broadcast_to(np.asarray(high, self.dtype), self.shape).copy()
            self._rng = np.random.default_rng()

        def sample(self):
            return self._rng.uniform(self.low, self.high).astype(self.dtype)

        def contains(self, x) -> bool:
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    class Dict:  # type: ignore
        def __init__(self, spaces):
            self.spaces = dict(spaces)

        def __getitem__(self, k):
            return self.spaces[k]

        def keys(self):
            return self.spaces.keys()

        def items(self):
            return self.spaces.items()

        def sample(self):
            return {k: s.sample() for k, s in self.spaces.items()}

        def contains(self, x) -> bool:
            return all(k in x and s.contains(x[k]) for k, s in self.spaces.items())

        def __repr__(self):
            return f"Dict({self.spaces})"


def action_space() -> "Box":
    """direction xyz in [-1,1], magnitude in [0,1] (exp03_vFinal_environment.py:191-198)."""
    return Box(low=np.array([-1, -1, -1, 0], np.float32), high=np.array([1, 1, 1, 1], np.float32), shape=(4,),
               dtype=np.float32)


def observation_space(lidar_shape=(3, 13, 26)) -> "Dict":
    """Dict{lidar [C,13,26] in [0,1]; inertial_data [15] in [-1,1]; last_action [4]}
    (exp03_vFinal_environment.py:230-278)."""
    return Dict({
        "lidar": Box(0, 1, shape=tuple(lidar_shape), dtype=np.float32),
        "inertial_data": Box(-np.ones(15, np.float32), np.ones(15, np.float32), shape=(15,), dtype=np.float32),
        "last_action": Box(np.array([-1, -1, -1, 0], np.float32), np.array([1, 1, 1, 1], np.float32), shape=(4,),
                           dtype=np.float32),
    })


def stacked_observation_space(n_spheres=6, sphere_shape=(3, 13, 26)) -> "Dict":
    """level5 student observation: Dict{stacked_spheres [6,C,13,26] in [0,1]; validity_mask [6] in {0,1};
    inertial_data [15]; last_action [4]} (threatsense/level5/level5_envrionment.py:312-351,380-440)."""
    return Dict({
        "stacked_spheres": Box(0, 1, shape=(n_spheres, *sphere_shape), dtype=np.float32),
        "validity_mask": Box(0, 1, shape=(n_spheres,), dtype=np.uint8),
        "inertial_data": Box(-np.ones(15, np.float32), np.ones(15, np.float32), shape=(15,), dtype=np.float32),
        "last_action": Box(np.array([-1, -1, -1, 0], np.float32), np.array([1, 1, 1, 1], np.float32), shape=(4,),
                           dtype=np.float32),
    })
