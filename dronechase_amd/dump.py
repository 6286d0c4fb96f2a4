"""On-disk dump of level5 student observations and teacher actions in the layout of the reference's collector
(apps/threatsense_runner/collect_and_save.py:52-97): datasets

    student/stacked_spheres [M,6,3,13,26] f32   student/validity_mask [M,6] bool
    student/inertial_data   [M,15] f32          student/last_action   [M,4] f32
    teacher_actions         [M,4] f32

one row per kept observation; rows whose validity mask is all False are dropped as `drop_invalid_student_obs` does
(:100-112).  The reference appends row by row through h5py; here whole device batches are appended.  h5py is not part
of this image: when it is importable the files are HDF5 (`partK.h5`, resizable chunked datasets, same names), otherwise
each part is a NumPy `.npz` whose keys are the same dataset paths -- readable with `numpy.load` and convertible with
`to_hdf5` wherever h5py exists.  SURVEY.md 8(f) item 4."""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import numpy as np

STUDENT_KEYS = ("stacked_spheres", "validity_mask", "inertial_data", "last_action")


def _have_h5py() -> bool:
    try:
        import h5py  # noqa: F401
        return True
    except Exception:
        return False


def _rows(x) -> np.ndarray:
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


class ObservationDump:
    """`add(obs, teacher_actions)` with batched arrays/tensors; a part file is written every `rows_per_file` kept rows
    (the reference flushes at >= 1000, collect_and_save.py:186-201) and on `close()`."""

    def __init__(self, directory: str, rows_per_file: int = 1000, use_hdf5: Optional[bool] = None):
        self.directory, self.rows_per_file = directory, int(rows_per_file)
        self.use_hdf5 = _have_h5py() if use_hdf5 is None else bool(use_hdf5)
        if self.use_hdf5 and not _have_h5py():
            raise ImportError("ObservationDump(use_hdf5=True): h5py is not importable here")
        os.makedirs(directory, exist_ok=True)
        self._pending: Dict[str, List[np.ndarray]] = {k: [] for k in (*STUDENT_KEYS, "teacher_actions")}
        self._n_pending = 0
        self.files: List[str] = []
        self.rows_written = 0

    def add(self, obs: Dict[str, object], teacher_actions) -> int:
        """Returns the number of rows kept from this batch."""
        mask = _rows(obs["validity_mask"]).astype(bool)
        keep = mask.any(axis=1)
        if not keep.any():
            return 0
        for k in STUDENT_KEYS:
            v = mask if k == "validity_mask" else _rows(obs[k]).astype(np.float32, copy=False)
            self._pending[k].append(v[keep])
        self._pending["teacher_actions"].append(_rows(teacher_actions).astype(np.float32, copy=False)[keep])
        self._n_pending += int(keep.sum())
        while self._n_pending >= self.rows_per_file:
            self._flush(self.rows_per_file)
        return int(keep.sum())

    def _flush(self, n: int) -> None:
        cat = {k: np.concatenate(v) for k, v in self._pending.items()}
        part = {k: v[:n] for k, v in cat.items()}
        self._pending = {k: [v[n:]] if len(v) > n else [] for k, v in cat.items()}
        self._n_pending -= n
        path = os.path.join(self.directory, f"part{len(self.files)}." + ("h5" if self.use_hdf5 else "npz"))
        if self.use_hdf5:
            write_hdf5(path, part)
        else:
            np.savez(path, **{("teacher_actions" if k == "teacher_actions" else f"student/{k}"): v for k, v in part.items()})
        self.files.append(path)
        self.rows_written += n

    def close(self) -> None:
        if self._n_pending:
            self._flush(self._n_pending)


def write_hdf5(path: str, part: Dict[str, np.ndarray]) -> None:
    """The reference's file layout: group `student`, resizable chunked datasets (collect_and_save.py:58-97)."""
    import h5py

    with h5py.File(path, "a") as f:
        g = f.require_group("student")
        for k, v in part.items():
            where, name = (f, k) if k == "teacher_actions" else (g, k)
            if name in where:
                d = where[name]
                d.resize(d.shape[0] + len(v), axis=0)
                d[-len(v):] = v
            else:
                where.create_dataset(name, data=v, maxshape=(None,) + v.shape[1:], chunks=True)


def load_part(path: str) -> Dict[str, np.ndarray]:
    """{"student/<key>": array, "teacher_actions": array} of one part file, either format."""
    if path.endswith(".npz"):
        with np.load(path) as z:
            return {k: z[k] for k in z.files}
    import h5py

    with h5py.File(path, "r") as f:
        out = {f"student/{k}": f["student"][k][...] for k in f["student"]}
        out["teacher_actions"] = f["teacher_actions"][...]
        return out


def to_hdf5(npz_path: str, h5_path: str) -> None:
    part = load_part(npz_path)
    write_hdf5(h5_path, {k.split("/", 1)[-1]: v for k, v in part.items()})
