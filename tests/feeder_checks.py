"""The product's feeder / augmentation functions against the fixtures of tests/golden/feeder.npz (outputs of the
literal NumPy oracle oracle/feeder_oracle.py on seeded inputs) -- one body for the host (NumPy / torch-CPU) and the
device (cuda) pipeline.  The random draws of the fixtures (epoch permutations from RandomState streams, rotation
angles, jitter noise) are fed to the product functions, so what is compared is the arithmetic."""
import numpy as np
import torch

from conftest import load_golden  # noqa: F401  (kept for symmetry with the other golden tests)


def fixtures():
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "feeder.npz"))
    return {k: z[k] for k in z.files}


def _close(got, want, scale=1.0):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    # fp32 results of O(scale) values: a few ulp of the largest intermediate
    np.testing.assert_allclose(got.astype(np.float64), np.asarray(want, dtype=np.float64), rtol=0, atol=4e-7 * scale)


def check_pipeline(dev):
    """dev: 'cpu' or 'cuda:0' -- where the torch side of the product's pipeline runs"""
    from scanobjectnn_amd import data_utils as DU
    from scanobjectnn_amd import provider
    fx = fixtures()
    raw = torch.from_numpy(fx["raw"]).to(dev)
    big = float(np.abs(fx["raw"]).max())
    # centre / normalise once per loaded set (train.py:100-106): device formulas against the oracle's per-cloud loops
    centred = DU.center_data_device(raw)
    _close(centred, fx["centered"], scale=big)
    data = DU.normalize_data_device(centred)
    # a cloud far from the origin carries its centring error (ulps of the RAW coordinates) into the unit sphere:
    # per cloud, 4e-7 x (largest raw coordinate / radius of the centred cloud + 1)
    radius = np.sqrt((fx["centered"].astype(np.float64) ** 2).sum(-1)).max(axis=1)
    for k in range(fx["raw"].shape[0]):
        _close(data[k], fx["normalized"][k], scale=float(np.abs(fx["raw"][k]).max() / radius[k] + 1.0))
    assert abs(float((data * data).sum(dim=2).sqrt().amax(dim=1).sub(1.0).abs().max())) < 1e-6
    # epoch view: the two permutations of get_current_data_h5 as index vectors, gathered on the device
    rs = np.random.RandomState(77)
    idx_pts, idx = DU.epoch_indices(6, 96, 64, rs)
    norm = torch.from_numpy(fx["normalized"]).to(dev)            # from here on: the oracle's own input, bit for bit
    cur = norm[:, torch.from_numpy(idx_pts).to(dev), :][torch.from_numpy(idx).to(dev)]
    assert np.array_equal(cur.cpu().numpy(), fx["cur_h5_data"])
    assert np.array_equal(fx["labels"][idx], fx["cur_h5_labels"])
    # augmentation with the fixture's draws
    rot = provider.rotate_point_cloud(norm, angles=fx["angles"])
    _close(rot, fx["rotated"])
    jit = provider.jitter_point_cloud(torch.from_numpy(fx["rotated"]).to(dev), noise=fx["noise"])
    _close(jit, fx["jittered"])
    d = (jit - torch.from_numpy(fx["rotated"]).to(dev)).abs().max().item()
    assert abs(d - 0.05) < 1e-6                                   # the planted +-7 / -9 sigma draws are clipped
    for i in range(4):
        out = provider.rotate_point_cloud_by_angle(norm, float(fx["vote_angle_%d" % i]))
        _close(out, fx["vote_rotated_%d" % i])
    # one whole epoch as the trainer composes it, the RandomState stream consumed in the reference's order
    rs = np.random.RandomState(82)
    idx_pts, idx = DU.epoch_indices(6, 96, 64, rs)
    whole = DU.normalize_data_device(DU.center_data_device(raw))
    cur = whole[:, torch.from_numpy(idx_pts).to(dev), :][torch.from_numpy(idx).to(dev)]
    for b in range(3):
        batch = cur[2 * b:2 * b + 2]
        angles = np.array([rs.uniform() * 2 * np.pi for _ in range(2)])
        batch = provider.rotate_point_cloud(batch, angles=angles)
        batch = provider.jitter_point_cloud(batch, noise=rs.randn(2, 64, 3))
        _close(batch, fx["epoch_batches"][b], scale=float((np.abs(fx["raw"]).max(axis=(1, 2)) / radius).max() + 1.0))
        assert np.array_equal(fx["labels"][idx][2 * b:2 * b + 2], fx["epoch_labels"][b])
