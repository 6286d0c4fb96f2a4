"""dronechase_amd.dump: the collector's on-disk layout (apps/threatsense_runner/collect_and_save.py:52-112).  h5py is not in
this image, so the .npz fallback (same dataset paths) is what runs here; the HDF5 writer is exercised where h5py exists."""
import numpy as np
import pytest

from dronechase_amd import dump


def _batch(n, seed, all_invalid=()):
    g = np.random.default_rng(seed)
    mask = g.random((n, 6)) < 0.4
    mask[:, 0] |= True
    for i in all_invalid:
        mask[i] = False
    obs = {"stacked_spheres": g.random((n, 6, 3, 13, 26), dtype=np.float32), "validity_mask": mask,
           "inertial_data": g.random((n, 15), dtype=np.float32), "last_action": g.random((n, 4), dtype=np.float32)}
    return obs, g.random((n, 4), dtype=np.float32)


def test_parts_keep_the_reference_layout_and_drop_invalid_rows(tmp_path):
    d = dump.ObservationDump(str(tmp_path), rows_per_file=100, use_hdf5=False)
    kept, all_obs, all_act = 0, [], []
    for s in range(5):
        obs, act = _batch(64, s, all_invalid=(3, 17))
        assert d.add(obs, act) == 62                      # drop_invalid_student_obs (:100-112)
        keep = obs["validity_mask"].any(1)
        all_obs.append({k: v[keep] for k, v in obs.items()}); all_act.append(act[keep]); kept += 62
    d.close()
    assert d.rows_written == kept == 310 and len(d.files) == 4 and [f.endswith(".npz") for f in d.files] == [True] * 4
    parts = [dump.load_part(f) for f in d.files]
    assert set(parts[0]) == {"student/stacked_spheres", "student/validity_mask", "student/inertial_data", "student/last_action", "teacher_actions"}
    assert [len(p["teacher_actions"]) for p in parts] == [100, 100, 100, 10]
    assert parts[0]["student/validity_mask"].dtype == np.bool_ and parts[0]["student/stacked_spheres"].dtype == np.float32
    assert parts[0]["student/stacked_spheres"].shape[1:] == (6, 3, 13, 26)
    for k in dump.STUDENT_KEYS:                            # order and content survive the re-chunking
        np.testing.assert_array_equal(np.concatenate([p[f"student/{k}"] for p in parts]), np.concatenate([o[k] for o in all_obs]))
    np.testing.assert_array_equal(np.concatenate([p["teacher_actions"] for p in parts]), np.concatenate(all_act))
    obs, act = _batch(4, 9, all_invalid=(0, 1, 2, 3))
    assert dump.ObservationDump(str(tmp_path / "x"), use_hdf5=False).add(obs, act) == 0


def test_hdf5_writer_when_h5py_is_importable(tmp_path):
    h5py = pytest.importorskip("h5py")
    d = dump.ObservationDump(str(tmp_path), rows_per_file=50)
    obs, act = _batch(80, 1)
    d.add(obs, act); d.close()
    with h5py.File(d.files[0], "r") as f:
        assert set(f["student"]) == set(dump.STUDENT_KEYS) and f["teacher_actions"].shape == (50, 4)
        assert f["student"]["stacked_spheres"].maxshape[0] is None
