"""the DEVICE-side input pipeline (centre / normalise / epoch view / rotate / jitter on
cuda tensors, scanobjectnn_amd.data_utils + provider) against the fixtures of the feeder oracle
(tests/golden/feeder.npz, oracle/feeder_oracle.py) -- not against the package's own NumPy functions."""
import pytest

import feeder_checks as FC

pytestmark = pytest.mark.gpu


def test_device_pipeline_against_the_fixtures():
    FC.check_pipeline("cuda:0")


def test_device_random_draws_have_the_reference_distributions():
    """the draws themselves come from torch generators on the device (the reference: NumPy's global stream): uniform
    angles in [0, 2 pi) per cloud, N(0, sigma) noise clipped at +-clip"""
    import math
    import torch
    from scanobjectnn_amd import provider
    g = torch.Generator(device="cuda:0")
    g.manual_seed(5)
    x = torch.zeros((4096, 8, 3), device="cuda:0")
    x[..., 0] = 1.0
    r = provider.rotate_point_cloud(x, generator=g)
    ang = torch.atan2(-r[:, 0, 2], r[:, 0, 0]) % (2 * math.pi)        # (1,0,0) -> (cos, 0, sin): angle recovered
    assert abs(ang.mean().item() - math.pi) < 0.15 and ang.min().item() >= 0 and ang.max().item() < 2 * math.pi
    assert abs(ang.std().item() - 2 * math.pi / math.sqrt(12)) < 0.1
    j = provider.jitter_point_cloud(torch.zeros((64, 2048, 3), device="cuda:0"), sigma=0.01, clip=0.02, generator=g)
    assert j.abs().max().item() <= 0.02 + 1e-9 and abs(j.std().item() - 0.01 * 0.88) < 1e-3   # std of N(0,1) clipped at 2
