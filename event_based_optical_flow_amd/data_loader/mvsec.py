"""MVSEC sequence reader (role of src/data_loader/mvsec.py:15-304).

On disk (mvsec.py:15-51, 98-113): `<root>/<sequence>_data.hdf5` with `davis/left/events` [N,4] = (x column, y row,
t seconds, p = -1/+1) and `davis/left/image_raw_ts` [F]; rectification maps `<root>/<sequence minus its last
character>_left_{x,y}_map.txt` (H lines of W numbers); ground truth `<gt>/<sequence>_gt_flow_dist.npz` with
`timestamps`, `x_flow_dist`, `y_flow_dist`.  Events are handed out as float64 [n,4] = (x ROW, y COLUMN, t, p)
(mvsec.py:178-207) -- the layout of `solver.optimize(events)`.

HDF5 is read through h5py when it is installed (it is not part of this image); a sequence exported once with
`numpy.savez(<sequence>_data.npz, **{"davis/left/events": ..., "davis/left/image_raw_ts": ...})` is read without it.
"""
import logging
import os

import numpy as np

from .base import DataLoaderBase

logger = logging.getLogger(__name__)

EVENTS_KEY = "davis/left/events"
GRAY_TS_KEY = "davis/left/image_raw_ts"

# valid ground-truth frames per sequence (mvsec.py:127-146): (first, last) with Python slice semantics
VALID_GT_FRAMES = {"indoor_flying1": (60, 1340), "indoor_flying2": (140, 1500), "indoor_flying3": (100, 1711),
                   "indoor_flying4": (104, 380), "outdoor_day1": (0, 5020), "outdoor_day2": (30, -1)}


def read_event_file(path: str):
    """-> (events int16 [N,4] (x col, y row, t truncated, p), timestamps float64 [N], gray_ts float64 [F]).
    The int16 copy of the event array is what the reference keeps in memory (mvsec.py:26-34); the timestamps
    come from column 2 of the un-cast dataset (mvsec.py:48-50)."""
    if path.endswith(".npz"):
        with np.load(path) as f:
            raw, gray = np.asarray(f[EVENTS_KEY]), np.asarray(f[GRAY_TS_KEY], dtype=np.float64)
    else:
        try:
            import h5py
        except ImportError as e:
            raise ImportError(f"reading {path} needs h5py; alternatively export the sequence to "
                              f"{os.path.splitext(path)[0]}.npz (keys '{EVENTS_KEY}', '{GRAY_TS_KEY}')") from e
        with h5py.File(path, "r") as f:
            raw = np.asarray(f["davis"]["left"]["events"])
            gray = np.asarray(f["davis"]["left"]["image_raw_ts"], dtype=np.float64)
    return raw.astype(np.int16), np.array(raw[:, 2], dtype=np.float64), gray


def undistort_events(events: np.ndarray, map_x: np.ndarray, map_y: np.ndarray, h: int, w: int) -> np.ndarray:
    """Rectify (x row, y col) through the look-up maps; events mapped outside the sensor are dropped
    (src/utils/event_utils.py:91-115)."""
    r, c = events[:, 0].astype(np.int32), events[:, 1].astype(np.int32)
    k, l = np.int32(map_y[r, c]), np.int32(map_x[r, c])
    out = np.copy(events)
    out[:, 0], out[:, 1] = k, l
    return out[(0 <= k) & (k < h) & (0 <= l) & (l < w)]


def _sample_nearest(img: np.ndarray, x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """cv2.remap(img, x, y, INTER_NEAREST) with the default constant-0 border."""
    xi, yi = np.rint(x).astype(np.int64), np.rint(y).astype(np.int64)
    ok = (xi >= 0) & (xi < img.shape[1]) & (yi >= 0) & (yi < img.shape[0])
    out = np.zeros(x.shape, dtype=img.dtype)
    out[ok] = img[yi[ok], xi[ok]]
    return out


def estimate_corresponding_gt_flow(x_flow_in, y_flow_in, gt_timestamps, start_time, end_time):
    """Ground-truth pixel displacement between two timestamps: the displacement maps (one per ground-truth frame,
    each valid until the next) are chained by moving a grid of points through them; points that hit a zero-flow
    pixel are masked out (src/utils/flow_utils.py:763-857, after EV-FlowNet's evaluation code)."""
    it = int(np.searchsorted(gt_timestamps, start_time, side="right")) - 1
    gt_dt = gt_timestamps[it + 1] - gt_timestamps[it]
    x_flow, y_flow = np.squeeze(x_flow_in[it]), np.squeeze(y_flow_in[it])
    dt = end_time - start_time
    if gt_dt >= dt:  # inside one ground-truth interval: scale
        return x_flow * dt / gt_dt, y_flow * dt / gt_dt
    xs, ys = np.meshgrid(np.arange(x_flow.shape[1]), np.arange(x_flow.shape[0]))
    xs, ys = xs.astype(np.float32), ys.astype(np.float32)
    x0, y0 = xs.copy(), ys.copy()
    x_mask, y_mask = np.ones(xs.shape, dtype=bool), np.ones(ys.shape, dtype=bool)

    def advance(fx, fy, scale=1.0):
        sx, sy = _sample_nearest(fx, xs, ys), _sample_nearest(fy, xs, ys)
        x_mask[sx == 0] = False
        y_mask[sy == 0] = False
        xs[...] += sx * scale
        ys[...] += sy * scale

    advance(x_flow, y_flow, (gt_timestamps[it + 1] - start_time) / gt_dt)
    it += 1
    while gt_timestamps[it + 1] < end_time:
        advance(np.squeeze(x_flow_in[it]), np.squeeze(y_flow_in[it]))
        it += 1
    advance(np.squeeze(x_flow_in[it]), np.squeeze(y_flow_in[it]),
            (end_time - gt_timestamps[it]) / (gt_timestamps[it + 1] - gt_timestamps[it]))
    x_shift, y_shift = xs - x0, ys - y0
    x_shift[~x_mask] = 0
    y_shift[~y_mask] = 0
    return x_shift, y_shift


class MvsecDataLoader(DataLoaderBase):
    NAME = "MVSEC"

    def set_sequence(self, sequence_name: str, undistort: bool = False) -> None:
        self.sequence_name = sequence_name
        self.dataset_files = self.get_sequence(sequence_name)
        self.left_event, self.left_ts, self.left_gray_ts = read_event_file(self.dataset_files["event"])
        if self.gt_flow_available:
            self.setup_gt_flow(os.path.join(self.gt_flow_dir, sequence_name))
            self.omit_invalid_data(sequence_name)
        self.undistort = undistort
        if self.undistort:
            self.calib_map_x, self.calib_map_y = self.get_calib_map(self.dataset_files["calib_map_x"],
                                                                    self.dataset_files["calib_map_y"])
        self.min_ts, self.max_ts = self.left_ts.min(), self.left_ts.max()
        self.data_duration = self.max_ts - self.min_ts

    def get_sequence(self, sequence_name: str) -> dict:
        data_path = os.path.join(self.root_dir, sequence_name)
        event_file = data_path + "_data.hdf5"
        if not os.path.exists(event_file) and os.path.exists(data_path + "_data.npz"):
            event_file = data_path + "_data.npz"
        return {"event": event_file, "calib_map_x": data_path[:-1] + "_left_x_map.txt",
                "calib_map_y": data_path[:-1] + "_left_y_map.txt"}

    # -- ground truth -------------------------------------------------------------------------------------
    def setup_gt_flow(self, path: str):
        with np.load(path + "_gt_flow_dist.npz") as gt:
            self.gt_timestamps = np.asarray(gt["timestamps"])
            self.U_gt_all = np.asarray(gt["x_flow_dist"])
            self.V_gt_all = np.asarray(gt["y_flow_dist"])

    def free_up_flow(self):
        del self.gt_timestamps, self.U_gt_all, self.V_gt_all

    def omit_invalid_data(self, sequence_name: str):
        """Keep the valid ground-truth frames and the events between the first and the last of them."""
        first, last = 0, -1
        for key, span in VALID_GT_FRAMES.items():
            if key in sequence_name:
                first, last = span
                break
        self.gt_timestamps = self.gt_timestamps[first:last]
        self.U_gt_all = self.U_gt_all[first:last]
        self.V_gt_all = self.V_gt_all[first:last]
        i0, i1 = self.time_to_index(self.gt_timestamps[0]), self.time_to_index(self.gt_timestamps[-1])
        self.left_event, self.left_ts = self.left_event[i0:i1], self.left_ts[i0:i1]
        self.min_ts, self.max_ts = self.left_ts.min(), self.left_ts.max()
        self.left_gray_ts = self.left_gray_ts[(self.gt_timestamps[0] < self.left_gray_ts) & (self.gt_timestamps[-1] > self.left_gray_ts)]

    def gt_time_list(self):
        return self.gt_timestamps

    def eval_frame_time_list(self):
        return self.left_gray_ts  # MVSEC is evaluated on the grey-frame timestamps

    def get_gt_time(self, index: int) -> tuple:
        """(floor, ceil) ground-truth timestamps around event `index`; None where there is none."""
        inds = np.where(self.gt_timestamps > self.index_to_time(index))[0]
        if len(inds) == 0:
            return (self.gt_timestamps[-1], None)
        if len(inds) == len(self.gt_timestamps):
            return (None, self.gt_timestamps[0])
        return (self.gt_timestamps[inds[0] - 1], self.gt_timestamps[inds[0]])

    def load_optical_flow(self, t1: float, t2: float) -> np.ndarray:
        """Ground-truth pixel displacement between TIMESTAMPS t1 and t2: [H, W, 2], channels (row, column)."""
        u, v = estimate_corresponding_gt_flow(self.U_gt_all, self.V_gt_all, self.gt_timestamps, t1, t2)
        return np.stack((v, u), axis=2)

    # -- events -----------------------------------------------------------------------------------------
    def __len__(self):
        return len(self.left_event)

    def load_event(self, start_index: int, end_index: int, cam: str = "left") -> np.ndarray:
        """float64 [n,4] = (x row, y column, t absolute seconds, p in {-1, +1}); rectified when the sequence was set
        with undistort=True (events leaving the sensor are dropped)."""
        if cam != "left":
            raise NotImplementedError("only the left camera is read")
        if len(self.left_event) <= start_index:
            raise IndexError(f"events {start_index}..{end_index} requested, the sequence holds {len(self.left_event)}")
        sl = slice(start_index, end_index)
        events = np.zeros((end_index - start_index, 4), dtype=np.float64)
        events[:, 0] = self.left_event[sl, 1]
        events[:, 1] = self.left_event[sl, 0]
        events[:, 2] = self.left_ts[sl]
        events[:, 3] = self.left_event[sl, 3]
        if self.undistort:
            events = undistort_events(events, self.calib_map_x, self.calib_map_y, self._HEIGHT, self._WIDTH)
        return events

    def index_to_time(self, index: int) -> float:
        return self.left_ts[index]

    def time_to_index(self, time: float) -> int:
        return int(np.searchsorted(self.left_ts, time)) - 1

    # -- calibration -------------------------------------------------------------------------------------
    def load_calib(self) -> dict:
        """Intrinsics of the outdoor sequences (the only ones the reference hard-codes, mvsec.py:262-285)."""
        K = np.array([[223.9940010790056, 0, 170.7684322973841, 0], [0, 223.61783486959376, 128.18711828338436, 0],
                      [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
        return {"K": K}

    def get_calib_map(self, map_txt_x: str, map_txt_y: str):
        return self.load_map_txt(map_txt_x), self.load_map_txt(map_txt_y)

    def load_map_txt(self, map_txt: str) -> np.ndarray:
        out = np.zeros((self._HEIGHT, self._WIDTH))
        with open(map_txt, "r") as f:
            for i, line in enumerate(f.readlines()):
                out[i] = np.array([float(k) for k in line.split()])
        return out
