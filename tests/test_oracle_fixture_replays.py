"""The bodies of tests/test_gpu_fixtures.py (reference-made fixtures replayed as state blobs) run against the ORACLE on the CPU:
the same arenas, the same comparisons with the fixture, with the oracle standing where the C ABI stands on the GPU box.  Keeps the
replay code honest here (no GPU in this container) and pins the oracle's whole step — not only its unit entry points — on the
reference's numbers."""
import numpy as np
import pytest

import tests.test_gpu_fixtures as F


class _A:
    def __init__(self, a): self.a = np.array(a)
    def cpu(self): return self
    def numpy(self): return self.a


class _OracleAsEnv:
    def __init__(self, cfg):
        from oracle import te_oracle as O
        self.o, self.cfg = O.OracleEnv(cfg, "f32"), cfg
    def state_words(self): return self.o.state_words()
    def set_state(self, w): self.o.set_state(w)
    def get_state(self): return _A(self.o.get_state().view(np.int32))
    def step(self, a, terminal=False): return tuple(_A(x) for x in self.o.step(a, terminal=terminal))
    def observe(self): return tuple(_A(x) for x in self.o.observe())
    def close(self): self.o.close()


@pytest.fixture()
def on_oracle(monkeypatch):
    import dronechase_amd
    from oracle import te_oracle as O
    monkeypatch.setattr(F, "_gpu", lambda cfg: _OracleAsEnv(cfg))
    monkeypatch.setattr(F, "_load", lambda env, blob: env.set_state(blob.w))
    monkeypatch.setattr(F, "_zeros", lambda n: np.zeros((n, 4), np.float32))
    monkeypatch.setattr(dronechase_amd, "default_config", O.default_config)


@pytest.mark.parametrize("name", ["test_task_logic_fixture_through_the_c_abi", "test_lidar_binning_fixture_through_te_observe",
                                  "test_closer_wins_fixture_through_te_observe", "test_gun_fixture_through_the_c_abi",
                                  "test_normalization_fixture_through_te_observe"])
def test_replay_on_the_oracle(on_oracle, golden, name):
    getattr(F, name)(golden)


@pytest.mark.parametrize("variant", ["aco", "general"])
def test_kamikaze_replay_on_the_oracle(on_oracle, golden, variant):
    F.test_kamikaze_fixture_through_the_c_abi(golden, variant)
