"""Augmentation -- mirror of `pointnet2/utils/provider.py` (rotate_point_cloud :34-52, rotate_point_cloud_by_angle
:121-138, jitter_point_cloud :189-200, shuffle_points :22-32, random_scale / shift), vectorised and device
resident: the reference loops over clouds in Python on the host (`provider.py:43-51`) and feeds the result
through `feed_dict`; at >10k clouds/s that loop and the copy would dominate, so these run on whatever device the
batch lives on (torch tensors, optional generator).  NumPy arrays are accepted too (returned as NumPy).

Rotation convention of the reference: row vectors times [[c,0,s],[0,1,0],[-s,0,c]] (about the up axis Y).
"""
import math

import numpy as np
import torch


def _as_tensor(batch_data):
    if isinstance(batch_data, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(batch_data, dtype=np.float32)), True
    return batch_data, False


def _ret(t, was_numpy):
    return t.cpu().numpy() if was_numpy else t


def _rot_y(angles):
    """(B,) angles -> (B,3,3) matrices, reference layout"""
    c, s = torch.cos(angles), torch.sin(angles)
    z, o = torch.zeros_like(c), torch.ones_like(c)
    return torch.stack([torch.stack([c, z, s], -1), torch.stack([z, o, z], -1), torch.stack([-s, z, c], -1)], -2)


def rotate_point_cloud(batch_data, generator=None, angles=None):
    """random rotation about the up axis, one angle per cloud (provider.py:34-52).  angles: (B,) the per-cloud angles
    instead of fresh draws (the reference draws `np.random.uniform() * 2 * np.pi` per cloud, :45)"""
    x, np_in = _as_tensor(batch_data)
    if angles is not None:       # given angles: cos / sin in float64 like the reference's NumPy matrix (:46-50), then cast
        ang = torch.as_tensor(angles, dtype=torch.float64, device=x.device)
    else:
        ang = torch.rand(x.shape[0], generator=generator, device=x.device if generator is None or
                         generator.device == x.device else generator.device).to(x.device) * (2 * math.pi)
    out = torch.bmm(x[..., :3].reshape(x.shape[0], -1, 3), _rot_y(ang).to(x.dtype))
    return _ret(out, np_in)


def rotate_point_cloud_by_angle(batch_data, rotation_angle):
    """fixed rotation about the up axis (provider.py:121-138) -- the vote rotations of the evaluation loop"""
    x, np_in = _as_tensor(batch_data)
    ang = torch.full((1,), float(rotation_angle), dtype=torch.float64, device=x.device)   # float64 matrix (:130-134), cast
    out = x.clone()
    out[..., :3] = x[..., :3] @ _rot_y(ang)[0].to(x.dtype)
    return _ret(out, np_in)


def jitter_point_cloud(batch_data, sigma=0.01, clip=0.05, generator=None, noise=None):
    """per-point clipped Gaussian jitter (provider.py:189-200).  noise: the standard-normal draws (same shape) instead
    of fresh ones (the reference draws `np.random.randn(B, N, C)`, :198)"""
    assert clip > 0
    x, np_in = _as_tensor(batch_data)
    if noise is not None:
        noise = torch.as_tensor(noise, dtype=x.dtype, device=x.device)
    else:
        gdev = generator.device if generator is not None else x.device
        noise = torch.randn(x.shape, generator=generator, device=gdev, dtype=x.dtype).to(x.device)
    return _ret(x + torch.clamp(sigma * noise, -clip, clip), np_in)


def shuffle_points(batch_data, generator=None):
    """the same random point permutation for every cloud of the batch (provider.py:22-32)"""
    x, np_in = _as_tensor(batch_data)
    gdev = generator.device if generator is not None else x.device
    idx = torch.randperm(x.shape[1], generator=generator, device=gdev).to(x.device)
    return _ret(x[:, idx, :], np_in)


def random_scale_point_cloud(batch_data, scale_low=0.8, scale_high=1.25, generator=None):
    x, np_in = _as_tensor(batch_data)
    gdev = generator.device if generator is not None else x.device
    s = torch.rand(x.shape[0], 1, 1, generator=generator, device=gdev).to(x.device) * (scale_high - scale_low) + scale_low
    return _ret(x * s, np_in)


def shift_point_cloud(batch_data, shift_range=0.1, generator=None):
    x, np_in = _as_tensor(batch_data)
    gdev = generator.device if generator is not None else x.device
    s = (torch.rand(x.shape[0], 1, 3, generator=generator, device=gdev).to(x.device) * 2 - 1) * shift_range
    return _ret(x + s, np_in)
