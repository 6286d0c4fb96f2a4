"""substeps_kernel<..., HELP> (small shards: a second wave next to every flight computes the motor noise of all sub-steps and hands it
over through LDS) must fly exactly what the plain kernel flies: same Philox words, same Box-Muller, only computed by another wave.  Outputs of
every step and the state blob bitwise, for the three task families, both controller rates, mixed and dense waves, ragged N.  TE_K1_HELP=1 / 0
selects the variant when the te_env is created."""
import pytest

pytestmark = pytest.mark.gpu


def _pair(monkeypatch, task, n, **over):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    from dronechase_amd import default_config
    from dronechase_amd.batched_env import BatchedEnv
    envs = []
    for mode in ("1", "0"):
        monkeypatch.setenv("TE_K1_HELP", mode)
        envs.append(BatchedEnv(default_config(task, n_envs=n, motor_noise=1, **over), "cuda:0"))
    monkeypatch.delenv("TE_K1_HELP")
    return envs


@pytest.mark.parametrize("task,n,over", [
    ("stage03", 8192, {}), ("stage03", 1000, {"seed": 3, "max_step": 20}), ("stage03", 2048, {"control_every_substep": 0}),
    ("stage01", 4096, {}), ("stage01", 777, {"control_every_substep": 0}), ("stage02", 4096, {"n_invaders": 8}), ("exp02", 2048, {"quad_preset": 0}),
])
def test_helped_flights_equal_plain_flights(monkeypatch, task, n, over):
    import torch
    a, b = _pair(monkeypatch, task, n, **over)
    ra, rb = a.reset(), b.reset()
    for x, y in zip(ra, rb):
        assert torch.equal(x, y)
    for s in range(200):
        act = a.random_actions(5, s)
        oa, ob = a.step(act), b.step(act)
        for k, (x, y) in enumerate(zip(oa, ob)):
            assert torch.equal(x, y), f"output {k} differs at step {s}"
        if s % 10 == 9:
            assert torch.equal(a.get_state(), b.get_state()), f"state differs after step {s}"
    a.close(); b.close()


def test_helped_flights_with_the_persistent_observation(monkeypatch):
    """Erase waves (fill.mode 3) in front of helped flight blocks."""
    import torch
    a, b = _pair(monkeypatch, "stage03", 4096, seed=6, max_step=15)
    a.set_persistent_obs(True); b.set_persistent_obs(True)
    a.reset(); b.reset()
    for s in range(60):
        act = a.random_actions(2, s)
        oa, ob = a.step(act), b.step(act)
        for x, y in zip(oa, ob):
            assert torch.equal(x, y)
    assert torch.equal(a.get_state(), b.get_state())
    a.close(); b.close()
