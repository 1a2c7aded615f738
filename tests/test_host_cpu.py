"""Host logic of the training loop, pinned by closed-form values: lr / BN-decay
schedules, TF-flavoured Adam on flat buffers, the synthetic-cloud normalisation, and the TF1-style variable store."""
import math

import pytest

import numpy as np
import torch

from scanobjectnn_amd import train_util as TU
from scanobjectnn_amd.graph import Graph, constant_initializer, get_variable, variable_scope, xavier_initializer
from scanobjectnn_amd.synth import center_data, normalize_data, synth_clouds


def test_learning_rate_schedule():
    """pointnet2/train.py:116-124 at B=16: staircase 0.7 every 200000 samples, floor 1e-5"""
    assert TU.get_learning_rate(0, 16) == 1e-3
    assert TU.get_learning_rate(12499, 16) == 1e-3                      # 199 984 samples
    assert math.isclose(TU.get_learning_rate(12500, 16), 7e-4)         # 200 000 samples
    assert math.isclose(TU.get_learning_rate(25000, 16), 4.9e-4)
    assert TU.get_learning_rate(10 ** 6, 16) == 1e-5                   # 0.7^80 << 1e-5 -> clipped


def test_bn_decay_schedule():
    """pointnet2/train.py:126-134: min(0.99, 1 - 0.5*0.5^floor(step*B/200000))"""
    assert TU.get_bn_decay(0, 16) == 0.5
    assert TU.get_bn_decay(12499, 16) == 0.5
    assert TU.get_bn_decay(12500, 16) == 0.75
    assert TU.get_bn_decay(25000, 16) == 0.875
    assert TU.get_bn_decay(10 ** 6, 16) == 0.99


def test_tf_adam_one_and_two_steps_by_hand():
    """tf.train.AdamOptimizer: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+eps) (epsilon OUTSIDE)"""
    lin = torch.nn.Linear(1, 1, bias=True)
    with torch.no_grad():
        lin.weight.fill_(2.0)
        lin.bias.fill_(-1.0)
    fp = TU.FlatParams(lin)
    opt = TU.TFAdam(fp)
    # every parameter starts on a 16-byte boundary of the flat buffers (the gaps stay zero)
    A = TU.FlatParams.ALIGN
    assert fp.flat.numel() == 2 * A and lin.weight.data_ptr() == fp.flat.data_ptr()
    assert lin.bias.data_ptr() == fp.flat.data_ptr() + 4 * A

    class _View:            # the two scalars, wherever the flat layout puts them
        def __init__(self, t):
            self.t = t

        def copy_(self, v):
            self.t.zero_()
            self.t[0], self.t[A] = v[0], v[1]

    real_flat, real_grad = fp.flat, fp.grad
    g1 = torch.tensor([0.5, -0.25])
    _View(fp.grad).copy_(g1)
    opt.step(0.1)
    fp.flat = real_flat[[0, A]]          # compare the two live entries; the gaps must not have moved
    assert real_flat[1:A].abs().max() == 0 and real_flat[A + 1:].abs().max() == 0
    m, v = 0.1 * g1, 0.001 * g1 * g1
    lr_t = 0.1 * math.sqrt(1 - 0.999) / (1 - 0.9)
    want = torch.tensor([2.0, -1.0]) - lr_t * m / (v.sqrt() + 1e-8)
    assert torch.allclose(fp.flat, want, atol=1e-7)
    fp.flat = real_flat
    g2 = torch.tensor([-1.0, 0.125])
    _View(fp.grad).copy_(g2)
    opt.step(0.1)
    fp.flat = real_flat[[0, A]]
    m, v = 0.9 * m + 0.1 * g2, 0.999 * v + 0.001 * g2 * g2
    lr_t = 0.1 * math.sqrt(1 - 0.999 ** 2) / (1 - 0.9 ** 2)
    want = want - lr_t * m / (v.sqrt() + 1e-8)
    assert torch.allclose(fp.flat, want, atol=1e-7)
    assert torch.allclose(lin.weight.flatten(), want[:1]) and torch.allclose(lin.bias, want[1:])


def test_flat_params_gradients_accumulate_in_place():
    net = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.ReLU(), torch.nn.Linear(4, 2))
    fp = TU.FlatParams(net)
    x = torch.randn(5, 3)
    net(x).sum().backward()
    ref = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    # the bucket holds every gradient at its 16-byte aligned offset and zeros in the gaps
    assert all(p.grad.data_ptr() % 16 == fp.grad.data_ptr() % 16 for p in net.parameters())
    assert torch.isclose(ref.abs().sum(), fp.grad.abs().sum()) and fp.grad.abs().sum() > 0
    fp.zero_grad()
    assert all(p.grad.abs().sum() == 0 for p in net.parameters())


def test_flat_params_collect_matches_in_place_accumulation():
    """begin_step()/collect(): gradients handed over by autograd and packed with one concatenation are the same
    flat bucket the in-place path produces; an unreached parameter contributes zeros"""
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.ReLU(), torch.nn.Linear(4, 2))
    extra = torch.nn.Parameter(torch.randn(7))          # never touched by the loss
    net.register_parameter("unused", extra)
    fp = TU.FlatParams(net)
    x = torch.randn(5, 3)
    fp.zero_grad()
    net(x).sum().backward()
    want = fp.grad.clone()
    fp.begin_step()
    assert all(p.grad is None for p in net.parameters())
    net(x).sum().backward()
    got = fp.collect()
    assert got is fp.grad and torch.equal(got, want)
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(fp.params, fp._gviews))
    assert extra.grad.abs().sum() == 0 and extra.grad.numel() == 7


def test_center_and_normalize():
    """data_utils.py:133-143,162-168"""
    rng = np.random.default_rng(0)
    pcs = rng.standard_normal((3, 50, 3)).astype(np.float32) * 4 + 2
    c = center_data(pcs)
    assert np.abs(c.mean(axis=1)).max() < 1e-5
    n = normalize_data(c)
    assert np.allclose(np.sqrt((n * n).sum(-1)).max(axis=1), 1.0, atol=1e-6)
    s = synth_clouds(4, 256, seed=1)
    assert s.dtype == np.float32 and s.shape == (4, 256, 3)
    assert np.allclose(np.sqrt((s * s).sum(-1)).max(axis=1), 1.0, atol=1e-5)
    assert np.array_equal(s, synth_clouds(4, 256, seed=1)) and not np.array_equal(s, synth_clouds(4, 256, seed=2))


def test_variable_store_scopes_and_reuse():
    g = Graph(seed=0)
    with g.as_default():
        with variable_scope("layer1"):
            with variable_scope("conv0"):
                w = get_variable("weights", [1, 1, 3, 64], xavier_initializer())
                b = get_variable("biases", [64], constant_initializer(0.0))
                with variable_scope("bn"):
                    mm = get_variable("moving_mean", [64], constant_initializer(0.0), trainable=False)
        with variable_scope("layer1"):
            with variable_scope("conv0"):
                w2 = get_variable("weights", [1, 1, 3, 64], xavier_initializer())
    assert w is w2
    names = dict(g.named_parameters())
    assert set(names) == {"layer1/conv0/weights", "layer1/conv0/biases"}
    assert "layer1/conv0/bn/moving_mean" in dict(g.named_buffers())
    limit = math.sqrt(6.0 / (3 + 64))
    assert w.abs().max().item() <= limit and w.abs().max().item() > 0.5 * limit and b.abs().sum() == 0
    assert not mm.requires_grad


def test_momentum_optimizer_matches_hand_computation():
    """tf.train.MomentumOptimizer (`pointnet2/train.py:165-166`): accum = m*accum + g; p -= lr*accum"""
    import torch
    from scanobjectnn_amd import train_util as TU
    lin = torch.nn.Linear(2, 1, bias=False)
    with torch.no_grad():
        lin.weight.copy_(torch.tensor([[1.0, -2.0]]))
    fp = TU.FlatParams(lin)
    opt = TU.make_optimizer("momentum", fp, 0.9)
    p, acc = np.array([1.0, -2.0]), np.zeros(2)
    for g in ([0.5, 0.25], [-1.0, 2.0], [0.125, 0.0]):
        lin.weight.grad.copy_(torch.tensor([g]))            # p.grad is a view of the flat gradient bucket
        opt.step(0.1)
        acc = 0.9 * acc + np.array(g)
        p = p - 0.1 * acc
    assert np.allclose(lin.weight.detach().numpy().ravel(), p, atol=1e-6)
    assert isinstance(TU.make_optimizer("adam", fp), TU.TFAdam)
    with pytest.raises(ValueError):
        TU.make_optimizer("sgd", fp)


def test_device_side_centre_normalise_equal_the_host_formulas():
    """data_utils.center_data_device / normalize_data_device (torch) == center_data / normalize_data (NumPy,
    `data_utils.py:133-143,162-168`) on un-normalised clouds"""
    import torch
    from scanobjectnn_amd import data_utils as DU
    rng = np.random.RandomState(2)
    pcs = (rng.randn(6, 50, 3) * [3.0, 0.5, 7.0] + [10.0, -4.0, 2.5]).astype(np.float32)
    want = DU.normalize_data(DU.center_data(pcs.copy()))
    got = DU.normalize_data_device(DU.center_data_device(torch.from_numpy(pcs))).numpy()
    assert np.allclose(got, want, atol=1e-6)
    assert np.allclose(np.sqrt((got ** 2).sum(-1)).max(1), 1.0, atol=1e-6)
    assert np.allclose(got.mean(1), 0.0, atol=1e-6)


def test_epoch_indices_reproduce_get_current_data_h5():
    from scanobjectnn_amd import data_utils as DU
    rng = np.random.RandomState(0)
    pcs = rng.rand(7, 30, 3).astype(np.float32)
    labels = np.arange(7)
    want, wl = DU.get_current_data_h5(pcs, labels, 12, rng=np.random.RandomState(5))
    ip, ic = DU.epoch_indices(7, 30, 12, np.random.RandomState(5))
    assert np.array_equal(pcs[:, ip][ic], want) and np.array_equal(labels[ic], wl)


def test_raw_object_loader(tmp_path):
    """`load_pc_file` / `load_data` (`data_utils.py:50-101`): float32 stream [count, 11 floats per point]; with_bg=False
    keeps the most frequent label outside {0,1,2}; clouds shorter than num_points are dropped"""
    import pickle
    from scanobjectnn_amd import data_utils as DU
    rng = np.random.RandomState(1)

    def write(name, n, labels):
        body = rng.rand(n, 11).astype(np.float32)
        body[:, 10] = labels          # the LAST column is what `load_pc_file` filters on (`pc[:,-1]`, :66-72)
        np.concatenate([[np.float32(n)], body.reshape(-1)]).astype(np.float32).tofile(tmp_path / name)
        return body
    a = write("a.bin", 40, np.r_[np.zeros(10), np.full(20, 5), np.full(10, 7)])
    write("b.bin", 8, np.full(8, 4))
    pc = DU.load_pc_file("a.bin", str(tmp_path))
    assert pc.shape == (40, 3) and np.array_equal(pc, a[:, :3])
    fg = DU.load_pc_file("a.bin", str(tmp_path), with_bg=False)
    assert fg.shape == (20, 3) and np.array_equal(fg, a[10:30, :3])
    with open(tmp_path / "split.pickle", "wb") as f:
        pickle.dump([{"filename": "objects_bin/a.bin", "label": 3}, {"filename": "objects_bin/b.bin", "label": 9}], f)
    pcs, labels = DU.load_data(str(tmp_path / "split.pickle"), num_points=16, data_path=str(tmp_path))
    assert len(pcs) == 1 and labels == [3]
    cur, lab = DU.get_current_data(pcs, labels, 16, rng=np.random.RandomState(0))
    assert cur.shape == (1, 16, 3) and lab.tolist() == [3]


def test_trainer_flags_are_the_references():
    """`pointnet2/train.py:25-46`: same flag names; the boolean switches are typed (the reference's are truthy strings)"""
    from scanobjectnn_amd.pointnet2 import train as T
    a = T.parse_args([])
    assert (a.with_bg, a.norm, a.center_data, a.num_class, a.optimizer, a.momentum) == (True, True, True, 15, "adam", 0.9)
    assert (a.num_point, a.batch_size, a.max_epoch, a.decay_step, a.decay_rate) == (1024, 16, 250, 200000, 0.7)
    b = T.parse_args(["--center_data", "false", "--norm", "0", "--optimizer", "momentum", "--momentum", "0.8",
                      "--num_class", "11", "--gpu", "0", "--normal", "--with_bg", "no"])
    assert (b.center_data, b.norm, b.with_bg, b.optimizer, b.momentum, b.num_class) == (False, False, False, "momentum", 0.8, 11)
