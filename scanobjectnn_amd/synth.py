"""Synthetic ScanObjectNN-shaped inputs.

Real clouds are centred and scaled to the unit sphere by the reference's feeder
(`data_utils.py:133-143` normalize_data, `:162-168` center_data); the generator
applies the same two steps so ball-query hit rates are realistic.  NumPy only.
"""
import numpy as np

NUM_CLASSES = 15  # training_data/shape_names_ext.txt


def center_data(pcs):
    """data_utils.py:162-168 -- subtract the per-cloud centroid."""
    pcs = np.asarray(pcs, dtype=np.float32)
    return pcs - pcs.mean(axis=1, keepdims=True, dtype=np.float32)


def normalize_data(pcs):
    """data_utils.py:133-143 -- divide by the per-cloud max L2 norm."""
    pcs = np.asarray(pcs, dtype=np.float32)
    d = np.sqrt((pcs * pcs).sum(axis=2, dtype=np.float32)).max(axis=1)
    return pcs / d[:, None, None]


def synth_clouds(B, N, seed=0, kind="surface"):
    """(B,N,3) float32 clouds.

    kind "ball":    uniform in the unit ball (worst case for ball query: expected
                    in-ball count ~ N r^3 < nsample, no early exit).
    kind "surface": ellipsoid shell + clipped jitter (`provider.py:189-200`), the
                    density profile of scanned objects.
    """
    rng = np.random.default_rng(seed)
    if kind == "ball":
        v = rng.standard_normal((B, N, 3))
        v /= np.linalg.norm(v, axis=2, keepdims=True)
        r = rng.random((B, N, 1)) ** (1.0 / 3.0)
        pts = v * r
    elif kind == "surface":
        v = rng.standard_normal((B, N, 3))
        v /= np.linalg.norm(v, axis=2, keepdims=True)
        axes = rng.uniform(0.3, 1.0, (B, 1, 3))
        pts = v * axes + np.clip(0.01 * rng.standard_normal((B, N, 3)), -0.05, 0.05)
    else:
        raise ValueError("kind must be 'ball' or 'surface'")
    pts = pts.astype(np.float32)
    return normalize_data(center_data(pts))


def synth_labels(B, seed=0):
    return np.random.default_rng(seed + 7919).integers(0, NUM_CLASSES, B).astype(np.int32)


def synth_masks(B, N, seed=0):
    return (np.random.default_rng(seed + 104729).random((B, N)) < 0.7).astype(np.int32)
