"""Parity AT THE BENCHMARKED SIZE (BASELINE config 2: 256 clouds x 2048 points): the code paths `bench.py` times --
persistent-workgroup tile loops over 4.19 M / 2.10 M rows, the 512-row-group statistics cap, 64-bit row offsets,
the arithmetic (never stored) first layer -- against torch float64 on the GPU, and the SSG logits of the full batch
against the chunked float64 CPU restatement."""
import numpy as np
import pytest
import torch

import mlp_ref as MR
from oracle import ref_models as R
from scanobjectnn_amd import fused_mlp
from scanobjectnn_amd.graph import Model
from scanobjectnn_amd.pointnet2 import tf_grouping, tf_sampling
from scanobjectnn_amd.synth import synth_clouds
from test_fused_mlp_gpu import _gather_backward_check, make_layers

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EPS = 1e-3


def _level_inputs(level):
    """the geometry of SA1 / SA2 of the SSG config on the bench's own synthetic clouds (real ball-query padding)"""
    x = torch.from_numpy(synth_clouds(256, 2048, seed=1234)).to(DEV)
    q1 = tf_sampling.gather_point(x, tf_sampling.farthest_point_sample(512, x))
    if level == "sa1":
        idx, _ = tf_grouping.query_ball_point(0.2, 32, x, q1)
        return x, q1, idx
    q2 = tf_sampling.gather_point(q1, tf_sampling.farthest_point_sample(128, q1))
    idx, _ = tf_grouping.query_ball_point(0.4, 64, q1, q2)
    return q1, q2, idx


@pytest.mark.parametrize("level", ["sa1", "sa2"])
def test_sa_stack_at_bench_size(level):
    """SA1: 4 194 304 grouped rows, coordinate-only first layer (xyz_bias form, never stored), widths 64-64-128.
    SA2: 2 097 152 rows, feature + coordinate first layer (q_xyz form), widths 128-128-256.  Forward (training and
    eval statistics) <= 1e-4 and every gradient against float64 autograd on the GPU."""
    xyz, new_xyz, idx = _level_inputs(level)
    g = torch.Generator().manual_seed(17)
    widths = [64, 64, 128] if level == "sa1" else [128, 128, 256]
    C1 = widths[0]
    B, N = xyz.shape[:2]
    src = {"Q": None if level == "sa1" else (0.5 * torch.randn(B, N, C1, generator=g)).to(DEV), "Ctr": None,
           "xyz": xyz, "new_xyz": new_xyz, "wxyz": torch.randn(3, C1, generator=g).to(DEV),
           "bias": (0.1 * torch.randn(C1, generator=g)).to(DEV) if level == "sa1" else None}
    layers = make_layers(C1, widths, seed=3)
    S = idx.shape[2]
    assert idx.numel() == (4194304 if level == "sa1" else 2097152)
    for training in (True, False):
        ls = [[t.clone() for t in l] for l in layers]
        out = fused_mlp.gather_mlp_stack(idx, True, training, 0.9, EPS, True, [tuple(l) for l in ls], Q=src["Q"],
                                         xyz=xyz, new_xyz=new_xyz, wxyz=src["wxyz"], bias=src["bias"])
        y1 = MR.gather_first_layer(src["Q"], None, xyz, new_xyz, src["wxyz"], src["bias"], idx, torch.float64)
        want = MR.run_stack(y1, None, layers, S, True, training, torch.float64)
        assert (out.double() - want).abs().max().item() < 1e-4
        del y1, want, out
        torch.cuda.empty_cache()
    _gather_backward_check(src, idx, layers, True)


def test_ssg_logits_at_bench_size_eval():
    """pointnet2_cls_ssg on the full (256, 2048, 3) batch, eval mode, against the float64 CPU restatement evaluated
    in chunks of 32 clouds (eval-mode BN makes clouds independent): |logit difference| <= 1e-4 for every cloud"""
    from test_models_parity_gpu import _randomise
    from scanobjectnn_amd.pointnet2 import pointnet2_cls_ssg as m
    c = synth_clouds(256, 2048, seed=1234)
    x = torch.from_numpy(c).to(DEV)
    net = Model(m.get_model, device=DEV, seed=1).build(x[:2].contiguous())
    _randomise(net, 5)
    P = R.params_from_state_dict(net.state_dict(), dtype=torch.float64)
    with torch.no_grad():
        logits, _ = net(x, is_training=False)
    logits = logits.cpu().double()
    torch.set_num_threads(min(32, torch.get_num_threads() or 1))
    worst = 0.0
    for lo in range(0, 256, 32):
        with torch.no_grad():
            want = R.pointnet2_cls_ssg(torch.from_numpy(c[lo:lo + 32]).double(), P, False)
        worst = max(worst, (logits[lo:lo + 32] - want).abs().max().item())
    assert worst <= 1e-4, worst
