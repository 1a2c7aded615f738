"""Generates tests/golden/feeder.npz: seeded inputs and the outputs of the feeder / augmentation ORACLE
(oracle/feeder_oracle.py -- a literal NumPy restatement; the reference's data_utils.py / provider.py cannot be imported
in this image: h5py, plyfile, pc_util, scipy.misc are absent).  Run from the repo root:

    python tests/golden/make_feeder_golden.py

The file holds DATA only (inputs, random draws, expected outputs).  The random draws come from
numpy.random.RandomState, the generator behind the reference's `np.random.*` calls, consumed in the reference's order.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import feeder_oracle as F  # noqa: E402


def make():
    rs = np.random.RandomState(2468)            # evaluate_seg_scenennobjects.py:79 uses this seed
    out = {}
    # raw clouds the way the h5 files hold them: not centred, not normalised, one flat cloud, one far from the origin
    raw = (rs.randn(6, 96, 3) * np.array([0.6, 1.3, 0.2]) + np.array([0.4, -2.0, 7.5])).astype(np.float32)
    raw[1] *= 40.0
    raw[2, :, 1] = 0.25
    labels = rs.randint(0, 15, 6).astype(np.int32)
    masks = rs.randint(-1, 4, (6, 96)).astype(np.int32)           # -1 = background, 0..3 = object part ids
    out["raw"], out["labels"], out["masks"] = raw, labels, masks
    out["centered"] = F.center_data(raw)
    out["normalized"] = F.normalize_data(out["centered"])
    out["binary_masks"] = F.convert_to_binary_mask(masks)
    data = out["normalized"]
    s = np.random.RandomState(77)
    out["cur_h5_data"], out["cur_h5_labels"] = F.get_current_data_h5(data, labels, 64, s)
    s = np.random.RandomState(78)
    out["cur_mask_data"], out["cur_mask_labels"], out["cur_mask_masks"] = \
        F.get_current_data_withmask_h5(data, labels, out["binary_masks"], 64, s, shuffle=True)
    out["cur_noshuf_data"], out["cur_noshuf_labels"], out["cur_noshuf_masks"] = \
        F.get_current_data_withmask_h5(data, labels, out["binary_masks"], 64, None, shuffle=False)
    s = np.random.RandomState(79)
    out["cur_parts_data"], out["cur_parts_labels"], out["cur_parts_parts"] = \
        F.get_current_data_parts_h5(data, labels, masks, 64, s)
    s = np.random.RandomState(80)
    out["shuffled_points"] = F.shuffle_points(data, s)
    s = np.random.RandomState(81)
    out["angles"] = F.rotation_angles(6, s)
    out["rotated"] = F.rotate_point_cloud(data, out["angles"])
    out["noise"] = F.jitter_noise(data.shape, s)
    out["noise"][0, 0] = [7.0, -9.0, 5.0]                         # beyond the clip on both sides
    out["jittered"] = F.jitter_point_cloud(out["rotated"], out["noise"])
    for i, a in enumerate((0.0, 2 * np.pi / 12, np.pi, 2 * np.pi * 11 / 12)):   # votes 0, 1, 6, 11 of 12
        out["vote_angle_%d" % i] = np.array(a)
        out["vote_rotated_%d" % i] = F.rotate_point_cloud_by_angle(data, a)
    # one training epoch as train.py composes it (centre, normalise, epoch subset / order, per batch rotate + jitter)
    s = np.random.RandomState(82)
    batches = F.train_batch_pipeline(raw, labels, 64, 2, s, augment=True)
    out["epoch_batches"] = np.stack([b for b, _ in batches])
    out["epoch_labels"] = np.stack([l for _, l in batches])
    return out


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "feeder.npz")
    np.savez_compressed(path, **make())
    print("wrote", path, os.path.getsize(path), "bytes")
