"""dronechase_amd — MI355X-native batched threat-engagement drone environments.

Only the hot path of DaviGuanabara/dronechase is here: env.step() of the stage01/02/03 environments as
hand-written HIP kernels behind a C ABI (include/threatengage.h), plus the Python mirror of the
reference's Gymnasium / SB3 VecEnv surface.  Importing this package never touches the GPU; creating an
environment does, and fails loudly when the HIP library or a GPU is missing."""
from . import config  # noqa: F401
from ._lib import TEError, default_config  # noqa: F401

__all__ = ["config", "default_config", "TEError"]
