"""Host-side data handling -- mirror of the reference's `data_utils.py` for the ScanObjectNN h5 files
(load_h5 / load_withmask_h5 :249-261, center_data :162-168, normalize_data :133-143, get_current_data_h5
:171-186, get_current_data_withmask_h5 :188-210, get_current_data_parts_h5 :212-229, load_parts_h5 :271-277,
convert_to_binary_mask :280-290).  NumPy only.

The image has no h5py: `load_h5*` import it lazily and say so if it is missing; the same arrays can be
shipped as `.npz` (`data`, `label`, `mask`) and read with `load_npz`.
"""
import numpy as np

from .synth import center_data, normalize_data  # noqa: F401  (same formulas; re-exported under the reference names)


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
