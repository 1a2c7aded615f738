"""Host-side data handling -- mirror of the reference's `data_utils.py` for the ScanObjectNN h5 files
(load_h5 / load_withmask_h5 :249-261, center_data :162-168, normalize_data :133-143, get_current_data_h5
:171-186, get_current_data_withmask_h5 :188-210, get_current_data_parts_h5 :212-229, load_parts_h5 :271-277,
convert_to_binary_mask :280-290).  NumPy only.

The image has no h5py: `load_h5*` import it lazily and say so if it is missing; the same arrays can be
shipped as `.npz` (`data`, `label`, `mask`) and read with `load_npz`.
"""
import os
import pickle

import numpy as np

from .synth import center_data, normalize_data  # noqa: F401  (same formulas; re-exported under the reference names)


# ---- device-side preprocessing ------------------------------------------------------------------
# The reference centres and normalises every loaded set once on the host (`pointnet2/train.py:100-106`); here the
# whole set is uploaded once (PB_T50_RS training: 11 416 x 2048 x 3 fp32 = 280 MB of the 288 GB) and the same two
# formulas run on the device.  Same arithmetic order as the NumPy versions above: fp32 mean over the points,
# sqrt of the fp32 sum of squares, one division.
def center_data_device(pcs):
    """torch (K,N,3): subtract the per-cloud centroid (data_utils.py:162-168)"""
    return pcs - pcs.mean(dim=1, keepdim=True)


def normalize_data_device(pcs):
    """torch (K,N,3): divide by the per-cloud max L2 norm (data_utils.py:133-143)"""
    d = (pcs * pcs).sum(dim=2).sqrt().amax(dim=1)
    return pcs / d[:, None, None]


def epoch_indices(num_clouds, num_total_points, num_points, rng):
    """the two permutations `get_current_data_h5` draws per epoch (:171-186): ONE point subset shared by every cloud,
    then the cloud order -- returned as index vectors so the gather can run on the device-resident set"""
    idx_pts = np.arange(num_total_points)
    rng.shuffle(idx_pts)
    idx = np.arange(num_clouds)
    rng.shuffle(idx)
    return idx_pts[:num_points], idx


# ---- raw object files (`load_pc_file` :50-75, `load_data` :77-101) --------------------------------------------------
def load_pc_file(filename, data_path="", suncg=False, with_bg=True):
    """one `.bin` object: float32 stream, entry 0 = point count, then 11 floats per point (x y z nx ny nz r g b
    label nyu_label) or 3 for SunCG.  with_bg=False keeps only the points of the most frequent non-background
    label (labels 0, 1, 2 are background classes)."""
    pc = np.fromfile(os.path.join(data_path, filename), dtype=np.float32)
    pc = pc[1:].reshape((-1, 3 if suncg else 11))
    if with_bg or suncg:
        return np.array(pc[:, 0:3])
    lab = pc[:, -1]
    keep = np.flatnonzero((lab != 0) & (lab != 1) & (lab != 2))
    values, counts = np.unique(lab[keep], return_counts=True)
    return np.array(pc[lab == values[np.argmax(counts)], 0:3])


def load_data(filename, num_points=1024, suncg_pl=False, with_bg_pl=True, data_path=""):
    """pickled list of {'filename', 'label'} -> (list of (n_i,3) arrays, list of labels); clouds with fewer than
    `num_points` points are dropped (:92-93)"""
    with open(filename, "rb") as handle:
        data = pickle.load(handle)
    pcs, labels = [], []
    for entry in data:
        pc = load_pc_file(entry["filename"].replace("objects_bin/", ""), data_path, suncg=suncg_pl, with_bg=with_bg_pl)
        if pc.shape[0] < num_points:
            continue
        pcs.append(pc)
        labels.append(entry["label"])
    return pcs, labels


def get_current_data(pcs, labels, num_points, rng=None):
    """ragged clouds: an independent random subset of `num_points` per cloud + a random cloud order (:108-131)"""
    rng = rng if rng is not None else np.random
    sampled = []
    for pc in pcs:
        if pc.shape[0] < num_points:
            raise ValueError("cloud with %d < %d points" % (pc.shape[0], num_points))
        idx = np.arange(pc.shape[0])
        rng.shuffle(idx)
        sampled.append(pc[idx[:num_points], :])
    sampled, labels = np.array(sampled), np.array(labels)
    idx = np.arange(len(labels))
    rng.shuffle(idx)
    return sampled[idx], labels[idx]


def _h5py():
    try:
        import h5py
        return h5py
    except ImportError as e:
        raise ImportError("h5py is not installed in this image: convert the ScanObjectNN .h5 file to .npz "
                          "(keys data/label[/mask]) and use data_utils.load_npz") from e


def load_h5(h5_filename):
    """data (K,2048,3) f32, label (K,) int"""
    f = _h5py().File(h5_filename, "r")
    return f["data"][:].astype(np.float32), f["label"][:].astype(np.int32)


def load_withmask_h5(h5_filename):
    f = _h5py().File(h5_filename, "r")
    return f["data"][:].astype(np.float32), f["label"][:].astype(np.int32), f["mask"][:].astype(np.int32)


def load_npz(filename, with_mask=False):
    z = np.load(filename)
    out = (z["data"].astype(np.float32), z["label"].astype(np.int32).reshape(-1))
    if with_mask == "parts":
        return out + (z["parts"].astype(np.int32),)
    return out + (z["mask"].astype(np.int32),) if with_mask else out


def get_current_data_h5(pcs, labels, num_points, rng=None):
    """ONE random point subset shared by every cloud of the epoch + a random cloud order (:171-186)"""
    rng = rng if rng is not None else np.random
    idx_pts = np.arange(pcs.shape[1])
    rng.shuffle(idx_pts)
    sampled = pcs[:, idx_pts[:num_points], :]
    idx = np.arange(len(labels))
    rng.shuffle(idx)
    return sampled[idx], labels[idx]


def get_current_data_withmask_h5(pcs, labels, masks, num_points, shuffle=True, rng=None):
    """(:188-210)"""
    rng = rng if rng is not None else np.random
    idx_pts = np.arange(pcs.shape[1])
    if shuffle:
        rng.shuffle(idx_pts)
    sampled, sampled_mask = pcs[:, idx_pts[:num_points], :], masks[:, idx_pts[:num_points]]
    idx = np.arange(len(labels))
    if shuffle:
        rng.shuffle(idx)
    return sampled[idx], labels[idx], sampled_mask[idx]


def load_parts_h5(h5_filename):
    """part-segmentation files: data, label, parts (per-point part id, 6 classes) (:271-277)"""
    f = _h5py().File(h5_filename, "r")
    return f["data"][:].astype(np.float32), f["label"][:].astype(np.int32), f["parts"][:].astype(np.int32)


def get_current_data_parts_h5(pcs, labels, parts, num_points, rng=None):
    """one random point subset for the epoch (clouds AND their part labels) + a random cloud order (:212-229)"""
    rng = rng if rng is not None else np.random
    idx_pts = np.arange(pcs.shape[1])
    rng.shuffle(idx_pts)
    sampled, sampled_parts = pcs[:, idx_pts[:num_points], :], parts[:, idx_pts[:num_points]]
    idx = np.arange(len(labels))
    rng.shuffle(idx)
    return sampled[idx], labels[idx], sampled_parts[idx]


def convert_to_binary_mask(masks):
    """background label -1 -> 0, every object part -> 1 (:280-290; training_data/README.md:13-14)"""
    return (np.asarray(masks) != -1).astype(np.int32)
