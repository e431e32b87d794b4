"""MVSEC reader on a synthetic sequence written to a temporary directory (CPU only).  Semantics after
src/data_loader/mvsec.py: column order of the event array, int16 storage with float64 timestamps, index <-> time,
rectification look-up, valid ground-truth frames, ground-truth displacement between two timestamps."""
import sys
import types

import numpy as np
import pytest

from event_based_optical_flow_amd import data_loader
from event_based_optical_flow_amd.data_loader import mvsec

H, W = 26, 34


def _write_sequence(root, name="outdoor_day2", n=5000, n_gt=80):
    rng = np.random.default_rng(3)
    t = np.sort(rng.uniform(100.0, 104.0, n))
    raw = np.stack([rng.integers(0, W, n), rng.integers(0, H, n), t, rng.choice([-1.0, 1.0], n)], axis=1)  # (x col, y row, t, p)
    gray = np.linspace(100.05, 103.95, 60)
    np.savez(root / f"{name}_data.npz", **{mvsec.EVENTS_KEY: raw, mvsec.GRAY_TS_KEY: gray})
    # rectification maps: shift one column right, one row down; the last column / row leave the sensor
    cols, rows = np.meshgrid(np.arange(W), np.arange(H))
    for tag, arr in (("x", cols + 1), ("y", rows + 1)):
        with open(root / f"{name[:-1]}_left_{tag}_map.txt", "w") as f:
            for line in arr:
                f.write(" ".join(f"{v:.1f}" for v in line) + "\n")
    gt_t = np.linspace(100.0, 104.0, n_gt)
    np.savez(root / f"{name}_gt_flow_dist.npz", timestamps=gt_t, x_flow_dist=np.full((n_gt, H, W), 2.0),
             y_flow_dist=np.full((n_gt, H, W), -1.0))
    return raw, gray, gt_t


def _loader(root, **extra):
    cfg = {"height": H, "width": W, "root": str(root), "dataset": "MVSEC"}
    cfg.update(extra)
    return data_loader.collections["MVSEC"](cfg)


def test_events_indices_and_rectification(tmp_path):
    raw, gray, _ = _write_sequence(tmp_path)
    dl = _loader(tmp_path)
    dl.set_sequence("outdoor_day2")
    assert len(dl) == len(raw) and dl.left_event.dtype == np.int16 and dl.left_ts.dtype == np.float64
    ev = dl.load_event(100, 600)
    assert ev.shape == (500, 4) and ev.dtype == np.float64
    np.testing.assert_array_equal(ev[:, 0], raw[100:600, 1])  # x = row
    np.testing.assert_array_equal(ev[:, 1], raw[100:600, 0])  # y = column
    np.testing.assert_array_equal(ev[:, 2], raw[100:600, 2])  # full-precision timestamps, not the int16 copy
    np.testing.assert_array_equal(ev[:, 3], raw[100:600, 3])
    assert dl.index_to_time(7) == raw[7, 2]
    i = dl.time_to_index(raw[1234, 2] + 1e-9)
    assert i == 1234
    assert dl.min_ts == raw[:, 2].min() and dl.data_duration == raw[:, 2].max() - raw[:, 2].min()
    np.testing.assert_array_equal(dl.eval_frame_time_list(), gray)
    with pytest.raises(IndexError):
        dl.load_event(len(raw), len(raw) + 10)
    with pytest.raises(NotImplementedError):
        dl.load_event(0, 10, cam="right")
    # rectified: +1 row, +1 column; events on the last row / column are dropped
    dl.set_sequence("outdoor_day2", undistort=True)
    und = dl.load_event(100, 600)
    keep = (raw[100:600, 1] + 1 < H) & (raw[100:600, 0] + 1 < W)
    np.testing.assert_array_equal(und[:, 0], raw[100:600, 1][keep] + 1)
    np.testing.assert_array_equal(und[:, 1], raw[100:600, 0][keep] + 1)


def test_ground_truth_frames_and_displacement(tmp_path):
    raw, gray, gt_t = _write_sequence(tmp_path)
    dl = _loader(tmp_path, load_gt_flow=True, gt=str(tmp_path))
    dl.set_sequence("outdoor_day2")
    # outdoor_day2: ground-truth frames [30:-1]; events restricted to that span
    np.testing.assert_array_equal(dl.gt_time_list(), gt_t[30:-1])
    assert dl.left_ts.min() >= gt_t[30] - 1e-2 and dl.left_ts.max() <= gt_t[-2]
    assert (dl.eval_frame_time_list() > gt_t[30]).all() and (dl.eval_frame_time_list() < gt_t[-2]).all()
    lo, hi = dl.get_gt_time(len(dl) // 2)
    assert lo <= dl.index_to_time(len(dl) // 2) < hi
    # shorter than one ground-truth interval: scaled copy of that frame's displacement
    step = gt_t[1] - gt_t[0]
    f = dl.load_optical_flow(gt_t[40] + 0.1 * step, gt_t[40] + 0.6 * step)
    assert f.shape == (H, W, 2)
    np.testing.assert_allclose(f[..., 0], -0.5)  # channel 0 = rows (y_flow_dist), channel 1 = columns
    np.testing.assert_allclose(f[..., 1], 1.0)
    # across 2.5 intervals: the displacement maps are chained (constant field -> 2.5 x one frame) for points that stay inside
    f = dl.load_optical_flow(gt_t[40] + 0.5 * step, gt_t[43])
    inner = f[6:-6, 8:-8]
    np.testing.assert_allclose(inner[..., 0], -2.5, atol=1e-5)
    np.testing.assert_allclose(inner[..., 1], 5.0, atol=1e-5)
    assert (f[:, -1, 1] == 0).all()  # points leaving the image sample the zero border and are masked


def test_hdf5_branch_and_missing_h5py(tmp_path, monkeypatch):
    raw, gray, _ = _write_sequence(tmp_path)
    (tmp_path / "outdoor_day2_data.hdf5").write_bytes(b"")  # the file name is what selects the branch
    dl = _loader(tmp_path)
    monkeypatch.setitem(sys.modules, "h5py", None)  # import h5py -> ImportError
    with pytest.raises(ImportError, match="npz"):
        dl.set_sequence("outdoor_day2")

    class FakeFile(dict):
        def __init__(self, path, mode):
            super().__init__(davis={"left": {"events": raw, "image_raw_ts": gray}})

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    monkeypatch.setitem(sys.modules, "h5py", types.SimpleNamespace(File=FakeFile))
    dl.set_sequence("outdoor_day2")
    assert dl.dataset_files["event"].endswith("_data.hdf5") and len(dl) == len(raw)
    np.testing.assert_array_equal(dl.load_event(0, 50)[:, 2], raw[:50, 2])


def test_base_class_contract():
    base = data_loader.DataLoaderBase({"height": 4, "width": 5, "root": "/tmp/x", "dataset": ""})
    assert base.dataset_dir.endswith("example") and not base.gt_flow_available and not base.auto_undistort
    with pytest.raises(NotImplementedError):
        base.load_event(0, 1)
    with pytest.raises(NotImplementedError):
        base.set_sequence("s")
