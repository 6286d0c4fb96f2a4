"""C-ABI library: loads without a GPU, exports every symbol include/threatengage.h declares, struct
layouts agree between C and ctypes, and the task constants are the reference's.  No compute calls."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from dronechase_amd import _lib
    from dronechase_amd.build import build_library
    build_library()  # hipcc cross-compiles gfx950 without a GPU
    return _lib.load()


def test_exports_every_declared_symbol(lib):
    from dronechase_amd import _lib
    header = open(os.path.join(ROOT, "include", "threatengage.h")).read()
    body = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(te_[a-z0-9_]+)\s*\(", body))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None
    from dronechase_amd import config as K
    assert lib.te_abi_version() == K.TE_ABI_VERSION == 5


def test_struct_layout_matches_c(lib):
    from dronechase_amd import config as K, default_config
    cfg = default_config("exp03")
    assert cfg.struct_size == C.sizeof(K.Config)  # te_config_default writes sizeof(te_config)
    # enum offsets of the state blob as the header states them
    header = open(os.path.join(ROOT, "include", "threatengage.h")).read()
    for name, val in K.D.items():
        assert re.search(rf"TE_D_{name} = {val}\b", header), name
    for name, val in K.E.items():
        assert re.search(rf"TE_E_{name} = {val}\b", header), name
    assert re.search(rf"TE_DRONE_WORDS = {K.DRONE_WORDS}\b", header) and re.search(rf"TE_ENV_WORDS = {K.ENV_WORDS}\b", header)


def test_task_constants_are_the_references(lib):
    from dronechase_amd import config as K, default_config
    e3 = default_config("exp03")
    # exp03_vFinal_task.py:88-112 ; calculate_rounds(2, 20) = 9
    assert (e3.n_pursuers, e3.n_invaders, e3.n_rounds, e3.munition, e3.max_step, e3.step_increment) == (2, 9, 9, 20, 300, 100)
    assert (e3.born_radius, e3.shoot_range, e3.cooldown_steps) == (6.0, 1.0, 60)
    assert abs(e3.explosion_range - 0.2) < 1e-7 and abs(e3.hit_prob - 0.9) < 1e-7
    assert e3.ally_policy == K.ALLY_BT and e3.dome_radius == 20 and e3.lidar_radius == 40
    assert e3.substeps == 16 and abs(e3.physics_dt - 1 / 240) < 1e-9 and abs(e3.control_dt - 1 / 120) < 1e-9
    assert abs(e3.max_speed - 10 / 3.6) < 1e-6  # quadcopter.py:590-600
    e2 = default_config("exp02")
    assert (e2.n_pursuers, e2.n_invaders, e2.n_rounds) == (1, 6, 6)  # calculate_rounds(1, 20) = 6
    e4 = default_config("exp04")
    assert e4.ally_policy == K.ALLY_FROZEN and e4.approach_bonus_gain == 10
    l5 = default_config("level5")
    # level5_task.py:76-98 ; calculate_max_rounds(6, 20, 12) = ceil((-23 + sqrt(23^2 + 8 * 121)) / 2) = 8
    assert (l5.n_pursuers, l5.n_invaders, l5.n_rounds, l5.munition, l5.max_step, l5.stacked_obs) == (6, 12, 8, 20, 300, 1)
    assert l5.ally_policy == K.ALLY_BT and l5.lidar_radius == 40 and e3.stacked_obs == 0
    s1 = default_config("stage01")
    assert (s1.n_pursuers, s1.n_invaders, s1.dome_radius, s1.lidar_radius, s1.munition, s1.max_step) == (2, 1, 10, 20, 0, 300)
    assert abs(s1.catch_distance - 0.4) < 1e-7
    s2 = default_config("stage02")
    assert (s2.n_pursuers, s2.n_invaders, s2.dome_radius, s2.munition, s2.max_step) == (2, 5, 8, 4, 600)
    with pytest.raises(ValueError):
        default_config(99)
    with pytest.raises(AttributeError):
        default_config("exp03", no_such_field=1)


def test_algorithmic_bytes(lib):
    from dronechase_amd import _lib, default_config
    # SURVEY.md 8(d) / BASELINE.md table
    assert _lib.algorithmic_bytes_per_env_step(default_config("stage01")) == 5292
    assert _lib.algorithmic_bytes_per_env_step(default_config("stage02", n_invaders=8)) == 7756
    assert _lib.algorithmic_bytes_per_env_step(default_config("stage03")) == 8108


def test_create_fails_loudly_without_gpu(lib):
    """No CPU fallback: without a HIP device te_create must return an error, not an environment."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from dronechase_amd import default_config
    h = C.c_void_p()
    rc = lib.te_create(C.byref(default_config("exp03", n_envs=4)), 0, C.byref(h))
    assert rc != 0 and not h.value
    assert lib.te_last_error()
    from dronechase_amd import TEError
    from dronechase_amd.batched_env import BatchedEnv
    with pytest.raises(TEError):
        BatchedEnv(default_config("exp03", n_envs=4), "cuda:0")
    with pytest.raises(TEError):
        BatchedEnv(default_config("exp03", n_envs=4), "cpu")


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under dronechase_amd/ may import or link it."""
    pkg = os.path.join(ROOT, "dronechase_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".c", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "te_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f


def test_every_source_of_the_library_is_a_build_dependency(lib):
    """A logic edit must never leave a stale libthreatengage.so behind (it ships to the GPU box as it is): every file under
    csrc/ and include/ is a dependency, and touching any of them makes needs_build() true."""
    from dronechase_amd import build as B
    names = {os.path.basename(d) for d in B.deps()}
    assert {"te_env.hip", "te_config.c", "te_device.hpp", "te_logic.hpp", "te_stacked.hpp", "threatengage.h"} <= names
    # every quoted include of the HIP translation unit is covered
    for src in ("te_env.hip", "te_logic.hpp", "te_stacked.hpp", "te_device.hpp"):
        for inc in re.findall(r'#include "([^"]+)"', open(os.path.join(B.CSRC, src)).read()):
            assert os.path.basename(inc) in names, (src, inc)
    assert not B.needs_build()
    lib_mtime = os.path.getmtime(B.LIB)
    for name in ("te_logic.hpp", "te_stacked.hpp"):
        path = os.path.join(B.CSRC, name)
        st = os.stat(path)
        try:
            os.utime(path, (lib_mtime + 10, lib_mtime + 10))
            assert B.needs_build(), name
        finally:
            os.utime(path, (st.st_atime, st.st_mtime))
    assert not B.needs_build()
