"""SpiderCNN on the op library: the kNN + grouping front end bit-exact against the oracle, the classifier's
logits against the float64 restatement (oracle/ref_models.spidercnn_cls_xyz), forward + backward smoke."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from oracle import ref_models as R
from scanobjectnn_amd.graph import Model
from scanobjectnn_amd.spidercnn import spidercnn_cls_xyz as m
from scanobjectnn_amd.spidercnn import tf_util as st
from scanobjectnn_amd.synth import synth_clouds, synth_labels

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_front_end_matches_oracle():
    """knn_point(20, xyz, xyz) through pairwise squared distances + the literal selection sort: indices bit-exact
    (self first, distance 0), deltas equal to the gathered differences"""
    c = synth_clouds(3, 300, seed=11)
    idx, delta = m.front_end(torch.from_numpy(c).to(DEV), 20)
    _, want_idx = O.knn_point(20, c, c)
    assert np.array_equal(idx.cpu().numpy(), want_idx)
    assert np.array_equal(idx[:, :, 0].cpu().numpy(), np.tile(np.arange(300, dtype=np.int32), (3, 1)))
    want_delta = O.group_point(c, want_idx) - c[:, :, None, :]
    assert np.array_equal(delta.cpu().numpy(), want_delta)


def _randomise(net, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in list(net.named_parameters()) + list(net.named_buffers()):
            if name.endswith("gamma"):
                p.copy_((0.5 + torch.rand(p.shape, generator=g)).to(p.device))
            elif name.endswith("beta") or name.endswith("taylor/biases"):
                p.copy_((0.2 * torch.randn(p.shape, generator=g)).to(p.device))
            elif name.endswith("moving_mean"):
                p.copy_((0.1 * torch.randn(p.shape, generator=g)).to(p.device))
            elif name.endswith("moving_variance"):
                p.copy_((0.5 + torch.rand(p.shape, generator=g)).to(p.device))


@pytest.mark.parametrize("training", [False, True])
def test_logits_against_float64_restatement(training, monkeypatch):
    ident = lambda inputs, is_training, scope, keep_prob=0.5, noise_shape=None: inputs  # noqa: E731
    monkeypatch.setattr(st, "dropout", ident)
    c = synth_clouds(16, 128, seed=12)
    x = torch.from_numpy(c).to(DEV)
    net = Model(m.get_model, device=DEV, seed=5).build(x)
    names = set(net.state_dict())
    assert {"graph.fanConv1/taylor/weight_xyz", "graph.fanConv3/taylor/conv/weights", "graph.fanConv4/taylor/conv/gn/gamma",
            "graph.fc1/bn/moving_mean", "graph.fc3/biases"} <= names
    assert tuple(net.state_dict()["graph.fanConv2/taylor/conv/weights"].shape) == (1, 20, 32 * 5, 64)
    _randomise(net, 13)
    P = R.params_from_state_dict(net.state_dict(), dtype=torch.float64)
    with torch.no_grad():
        logits = net(x, is_training=training, bn_decay=0.9)
        want = R.spidercnn_cls_xyz(torch.from_numpy(c).double(), P, training)
    assert logits.shape == (16, 15)
    assert (logits.cpu().double() - want).abs().max().item() <= 1e-4


def test_forward_backward():
    x = torch.from_numpy(synth_clouds(4, 256, seed=14)).to(DEV)
    y = torch.from_numpy(synth_labels(4, seed=14)).to(DEV)
    net = Model(m.get_model, device=DEV, seed=0).build(x)
    m.get_loss(net(x, is_training=True, bn_decay=0.5), y).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
