"""CPU restatement of the reference's data feeder and augmentation formulas (rows a25 / f1) --
TEST INFRASTRUCTURE ONLY (tests/, and the fixture generator tests/golden/make_feeder_golden.py).  Nothing under
scanobjectnn_amd/ may import this.

PARITY UNPINNED against the reference's own code: `data_utils.py` imports h5py / plyfile / pc_util / scipy.misc at
module level and `pointnet2/utils/provider.py` imports h5py, none of which exist in this image, so neither module can
be imported here and the reference holds no fixtures for them.  What follows is a LITERAL restatement -- the same
per-cloud Python loops, the same NumPy calls in the same order, the same dtypes (float32 clouds, float64 rotation
matrices and noise, results stored back into float32 where the reference does), the same consumption of the global
NumPy random stream (passed in as a `numpy.random.RandomState`, the generator behind `np.random.*`) -- each function
citing the reference lines it follows.  tests/golden/feeder.npz holds seeded inputs and these functions' outputs; the
product's host functions and its device pipeline are tested against THOSE, not against themselves.
"""
import numpy as np


def normalize_data(pcs):
    """data_utils.py:133-143: per cloud, d = max over the points of sqrt(sum |x|^2), pc /= d (in place, float32)"""
    pcs = np.array(pcs, dtype=np.float32, copy=True)
    for pc in pcs:
        d = max(np.sum(np.abs(pc) ** 2, axis=-1) ** (1. / 2))
        pc /= d
    return pcs


def center_data(pcs):
    """data_utils.py:162-168: per cloud, subtract the centroid column by column (in place, float32)"""
    pcs = np.array(pcs, dtype=np.float32, copy=True)
    for pc in pcs:
        centroid = np.mean(pc, axis=0)
        pc[:, 0] -= centroid[0]
        pc[:, 1] -= centroid[1]
        pc[:, 2] -= centroid[2]
    return pcs


def get_current_data_h5(pcs, labels, num_points, rs):
    """data_utils.py:171-186: ONE shuffled point subset shared by every cloud, then a shuffled cloud order -- two draws
    from the global stream, in this order"""
    idx_pts = np.arange(pcs.shape[1])
    rs.shuffle(idx_pts)
    sampled = pcs[:, idx_pts[:num_points], :]
    idx = np.arange(len(labels))
    rs.shuffle(idx)
    return sampled[idx], labels[idx]


def get_current_data_withmask_h5(pcs, labels, masks, num_points, rs, shuffle=True):
    """data_utils.py:188-210"""
    idx_pts = np.arange(pcs.shape[1])
    if shuffle:
        rs.shuffle(idx_pts)
    sampled = pcs[:, idx_pts[:num_points], :]
    sampled_mask = masks[:, idx_pts[:num_points]]
    idx = np.arange(len(labels))
    if shuffle:
        rs.shuffle(idx)
    return sampled[idx], labels[idx], sampled_mask[idx]


def get_current_data_parts_h5(pcs, labels, parts, num_points, rs):
    """data_utils.py:212-229"""
    idx_pts = np.arange(pcs.shape[1])
    rs.shuffle(idx_pts)
    sampled = pcs[:, idx_pts[:num_points], :]
    sampled_parts = parts[:, idx_pts[:num_points]]
    idx = np.arange(len(labels))
    rs.shuffle(idx)
    return sampled[idx], labels[idx], sampled_parts[idx]


def convert_to_binary_mask(masks):
    """data_utils.py:280-290: ones, zero where the mask is -1 (the reference returns float64)"""
    binary_masks = []
    for i in range(masks.shape[0]):
        binary_mask = np.ones(masks[i].shape)
        bg_idx = np.where(masks[i, :] == -1)
        binary_mask[bg_idx] = 0
        binary_masks.append(binary_mask)
    return np.array(binary_masks)


def shuffle_points(batch_data, rs):
    """provider.py:22-32: one permutation of the points for the whole batch"""
    idx = np.arange(batch_data.shape[1])
    rs.shuffle(idx)
    return batch_data[:, idx, :]


def rotation_angles(batch_size, rs):
    """the draws of provider.py:45: one `np.random.uniform() * 2 * np.pi` per cloud, in cloud order"""
    return np.array([rs.uniform() * 2 * np.pi for _ in range(batch_size)])


def rotate_point_cloud(batch_data, angles):
    """provider.py:34-52 with the per-cloud angles given (rotation about the up axis; float64 matrix, float32 result)"""
    rotated_data = np.zeros(batch_data.shape, dtype=np.float32)
    for k in range(batch_data.shape[0]):
        rotation_angle = angles[k]
        cosval = np.cos(rotation_angle)
        sinval = np.sin(rotation_angle)
        rotation_matrix = np.array([[cosval, 0, sinval],
                                    [0, 1, 0],
                                    [-sinval, 0, cosval]])
        shape_pc = batch_data[k, ...]
        rotated_data[k, ...] = np.dot(shape_pc.reshape((-1, 3)), rotation_matrix)
    return rotated_data


def rotate_point_cloud_by_angle(batch_data, rotation_angle):
    """provider.py:121-138 (the vote rotations of the evaluation loops)"""
    rotated_data = np.zeros(batch_data.shape, dtype=np.float32)
    for k in range(batch_data.shape[0]):
        cosval = np.cos(rotation_angle)
        sinval = np.sin(rotation_angle)
        rotation_matrix = np.array([[cosval, 0, sinval],
                                    [0, 1, 0],
                                    [-sinval, 0, cosval]])
        shape_pc = batch_data[k, :, 0:3]
        rotated_data[k, :, 0:3] = np.dot(shape_pc.reshape((-1, 3)), rotation_matrix)
    return rotated_data


def jitter_noise(shape, rs):
    """the draw of provider.py:198: `np.random.randn(B, N, C)` (float64)"""
    return rs.randn(*shape)


def jitter_point_cloud(batch_data, noise, sigma=0.01, clip=0.05):
    """provider.py:189-200 with the standard-normal draws given: clip(sigma * noise, -clip, clip) + data (float64)"""
    assert clip > 0
    jittered_data = np.clip(sigma * noise, -1 * clip, clip)
    jittered_data += batch_data
    return jittered_data


def train_batch_pipeline(pcs, labels, num_points, batch_size, rs, augment=True):
    """the feeder of one training epoch as `pointnet2/train.py:100-106,229-247` composes it: the set is centred and
    normalised once, every epoch draws its point subset and cloud order, every batch is rotated then jittered --
    -> list of (batch (B,n,3) float64 as fed, labels (B,))"""
    data = normalize_data(center_data(pcs))
    cur, lab = get_current_data_h5(data, labels, num_points, rs)
    out = []
    for b in range(cur.shape[0] // batch_size):
        batch = cur[b * batch_size:(b + 1) * batch_size]
        if augment:
            batch = rotate_point_cloud(batch, rotation_angles(batch.shape[0], rs))
            batch = jitter_point_cloud(batch, jitter_noise(batch.shape, rs))
        out.append((batch, lab[b * batch_size:(b + 1) * batch_size]))
    return out
