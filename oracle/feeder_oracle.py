"""CPU restatement of the reference's data-feeder and augmentation FORMULAS (rows a25 / f1) --
TEST INFRASTRUCTURE ONLY (tests/, and the fixture generator tests/golden/make_feeder_golden.py).  Nothing under
scanobjectnn_amd/ may import this.

PARITY UNPINNED against the reference's own code: `data_utils.py` imports h5py / plyfile / pc_util / scipy.misc at
module level and `pointnet2/utils/provider.py` imports h5py, none of which exist in this image, so neither module can
be imported here to produce fixtures and the reference holds none for them.  What follows states the same arithmetic in
this repository's own words, keeping everything that decides the NUMBERS: which dtype each step runs in (float32 clouds;
the rotation matrix and the jitter noise in float64, results narrowed to float32 where the reference stores them in a
float32 array), per-cloud evaluation (a float32 mean / max over one cloud at a time), and the ORDER in which the global
NumPy random stream is consumed (a `numpy.random.RandomState` -- the generator behind `np.random.*` -- is passed in).
Each function cites the reference lines whose result it reproduces.  tests/golden/feeder.npz holds seeded inputs and
these functions' outputs; the product's host functions and its device pipeline are tested against THOSE, not against
themselves.
"""
import numpy as np


def _own_f32(clouds):
    return np.array(clouds, dtype=np.float32, copy=True)


def normalize_data(pcs):
    """data_utils.py:133-143 -- every cloud divided by its largest point norm: squares, row sums and the square root in
    float32, one cloud at a time"""
    out = _own_f32(pcs)
    for k in range(out.shape[0]):
        radius = ((np.abs(out[k]) ** 2).sum(axis=-1) ** 0.5).max()
        out[k] /= radius
    return out


def center_data(pcs):
    """data_utils.py:162-168 -- the float32 centroid of every cloud taken off coordinate by coordinate"""
    out = _own_f32(pcs)
    for k in range(out.shape[0]):
        mid = out[k].mean(axis=0)
        for axis in range(3):
            out[k][:, axis] -= mid[axis]
    return out


def _epoch_permutations(num_total_points, num_clouds, rs, shuffle=True):
    """the two draws every get_current_data_*_h5 makes, in this order: a permutation of the POINT slots (one subset
    shared by all clouds of the epoch), then a permutation of the clouds"""
    slots = np.arange(num_total_points)
    order = np.arange(num_clouds)
    if shuffle:
        rs.shuffle(slots)
        rs.shuffle(order)
    return slots, order


def get_current_data_h5(pcs, labels, num_points, rs):
    """data_utils.py:171-186"""
    slots, order = _epoch_permutations(pcs.shape[1], len(labels), rs)
    return pcs[:, slots[:num_points], :][order], labels[order]


def get_current_data_withmask_h5(pcs, labels, masks, num_points, rs, shuffle=True):
    """data_utils.py:188-210 (shuffle=False: file order, the first num_points points -- the evaluation scripts)"""
    slots, order = _epoch_permutations(pcs.shape[1], len(labels), rs, shuffle)
    keep = slots[:num_points]
    return pcs[:, keep, :][order], labels[order], masks[:, keep][order]


def get_current_data_parts_h5(pcs, labels, parts, num_points, rs):
    """data_utils.py:212-229"""
    slots, order = _epoch_permutations(pcs.shape[1], len(labels), rs)
    keep = slots[:num_points]
    return pcs[:, keep, :][order], labels[order], parts[:, keep][order]


def convert_to_binary_mask(masks):
    """data_utils.py:280-290 -- 1.0 for object points, 0.0 where the stored label is -1 (float64, like the reference's
    np.ones)"""
    return np.where(np.asarray(masks) == -1, 0.0, 1.0)


def shuffle_points(batch_data, rs):
    """provider.py:22-32 -- one permutation of the point slots for the whole batch"""
    slots = np.arange(batch_data.shape[1])
    rs.shuffle(slots)
    return batch_data[:, slots, :]


def rotation_angles(batch_size, rs):
    """the draws of provider.py:45: one uniform() per cloud, in cloud order, scaled to [0, 2 pi)"""
    return np.array([rs.uniform() * 2 * np.pi for _ in range(batch_size)])


def _about_up_axis(angle):
    """float64 matrix the reference multiplies ROW vectors with (provider.py:46-50): x' = c x - s z, y' = y, z' = s x + c z"""
    c, s = np.cos(angle), np.sin(angle)
    rot = np.zeros((3, 3))
    rot[0, 0], rot[0, 2] = c, s
    rot[1, 1] = 1
    rot[2, 0], rot[2, 2] = -s, c
    return rot


def rotate_point_cloud(batch_data, angles):
    """provider.py:34-52 with the per-cloud angles given: float32 points times a float64 matrix, narrowed to float32"""
    out = np.zeros(batch_data.shape, dtype=np.float32)
    for k, angle in enumerate(angles):
        out[k] = batch_data[k].reshape(-1, 3) @ _about_up_axis(angle)
    return out


def rotate_point_cloud_by_angle(batch_data, rotation_angle):
    """provider.py:121-138 (the vote rotations of the evaluation loops): the first three channels rotated, float32"""
    out = np.zeros(batch_data.shape, dtype=np.float32)
    rot = _about_up_axis(rotation_angle)
    for k in range(batch_data.shape[0]):
        out[k, :, 0:3] = batch_data[k, :, 0:3].reshape(-1, 3) @ rot
    return out


def jitter_noise(shape, rs):
    """the draw of provider.py:198: standard normals of the batch's shape (float64)"""
    return rs.randn(*shape)


def jitter_point_cloud(batch_data, noise, sigma=0.01, clip=0.05):
    """provider.py:189-200 with the draws given: sigma * noise clipped to +-clip, added to the data (float64)"""
    assert clip > 0
    return np.clip(sigma * noise, -clip, clip) + batch_data


def train_batch_pipeline(pcs, labels, num_points, batch_size, rs, augment=True):
    """one training epoch's feed as `pointnet2/train.py:100-106,229-247` composes it: the set centred, then normalised,
    once; the epoch's point subset and cloud order; every whole batch rotated, then jittered
    -> list of (batch (B,n,3) as fed (float64 after the jitter), labels (B,))"""
    cur, lab = get_current_data_h5(normalize_data(center_data(pcs)), labels, num_points, rs)
    fed = []
    for lo in range(0, cur.shape[0] - batch_size + 1, batch_size):
        batch = cur[lo:lo + batch_size]
        if augment:
            batch = rotate_point_cloud(batch, rotation_angles(batch.shape[0], rs))
            batch = jitter_point_cloud(batch, jitter_noise(batch.shape, rs))
        fed.append((batch, lab[lo:lo + batch_size]))
    return fed
