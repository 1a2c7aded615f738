"""SURVEY 8f-2 on the GPU path + the evaluation caller (`pointnet2/evaluate_scenennobjects.py:27-44,141,152-231`):
a TensorFlow tensor bundle -- assembled by the INDEPENDENT code path of tests/indep_bundle.py, not by the in-tree
writer -- is restored into a product `graph.Model` on the GPU (SSG, and DGCNN with its ExponentialMovingAverage shadow
names), the logits are bit-identical to the state-dict route, and the evaluation command line runs from that file:
votes, `pred_label.txt`, `log_evaluate.txt`, the per-class table."""
import os

import numpy as np
import pytest
import torch

from indep_bundle import _indep_bundle
from scanobjectnn_amd.graph import Model
from scanobjectnn_amd.synth import synth_clouds, synth_labels

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _randomise(net, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in list(net.named_parameters()) + list(net.named_buffers()):
            if name.endswith("moving_variance"):
                p.copy_((0.5 + torch.rand(p.shape, generator=g)).to(p.device))
            elif not name.endswith("weights"):
                p.copy_((0.3 * torch.randn(p.shape, generator=g)).to(p.device) + (1.0 if name.endswith("gamma") else 0.0))


def _tf_names(sd, ema_shadow):
    """state-dict keys -> the variable names a checkpoint of the reference holds: the scope names as they are, and for
    DGCNN the moving statistics under the EMA shadow names (`dgcnn/utils/tf_util.py:484-506`)"""
    out = {}
    for k, v in sd.items():
        name = k[len("graph."):]
        if ema_shadow and name.endswith("/bn/moving_mean"):
            s = name[:-len("/bn/moving_mean")]
            name = "%s/bn/%s/bn/moments/Squeeze/ExponentialMovingAverage" % (s, s)
        elif ema_shadow and name.endswith("/bn/moving_variance"):
            s = name[:-len("/bn/moving_variance")]
            name = "%s/bn/%s/bn/moments/Squeeze_1/ExponentialMovingAverage" % (s, s)
        out[name] = v.detach().cpu().numpy()
    # what a training run leaves next to the model variables: optimizer slots and counters, to be ignored
    first = next(n for n in out if n.endswith("weights"))
    out[first + "/Adam"] = np.zeros_like(out[first])
    out[first + "/Adam_1"] = np.zeros_like(out[first])
    out["beta1_power"] = np.array(0.5, dtype=np.float32)
    out["global_step"] = np.array(1234, dtype=np.int64)
    return out


@pytest.mark.parametrize("name", ["pointnet2_cls_ssg", "dgcnn"])
def test_bundle_restore_on_gpu_matches_state_dict_route(name, tmp_path):
    import importlib
    from scanobjectnn_amd.pointnet2 import evaluate_scenennobjects as EV
    from scanobjectnn_amd.pointnet2.train import MODELS
    mod = importlib.import_module(MODELS[name])
    x = torch.from_numpy(synth_clouds(4, 512, seed=11)).to(DEV)
    src = Model(mod.get_model, device=DEV, seed=3).build(x)
    _randomise(src, 7)
    sd = {k: v.clone() for k, v in src.state_dict().items()}
    prefix = str(tmp_path / "model.ckpt")
    _indep_bundle(prefix, _tf_names(sd, ema_shadow=(name == "dgcnn")), nshards=2, entries_per_block=3)
    assert os.path.exists(prefix + ".index") and os.path.exists(prefix + ".data-00001-of-00002")

    via_bundle = Model(mod.get_model, device=DEV, seed=99).build(x)
    info = EV.restore(via_bundle, prefix)
    assert info["format"] == "tf-bundle" and info["loaded"] == len(sd) and not info["missing"] and not info["unexpected"]
    via_sd = Model(mod.get_model, device=DEV, seed=98).build(x)
    torch.save(sd, tmp_path / "model.pt")
    info2 = EV.restore(via_sd, str(tmp_path / "model.pt"))
    assert info2["format"] == "state-dict" and not info2["missing"]
    for k in sd:
        assert torch.equal(via_bundle.state_dict()[k], sd[k]), k
    with torch.no_grad():
        a = via_bundle(x, is_training=False)[0]
        b = via_sd(x, is_training=False)[0]
        c = src(x, is_training=False)[0]
    assert a.is_cuda and torch.equal(a, b) and torch.equal(a, c)
    with pytest.raises(FileNotFoundError):
        EV.restore(via_sd, str(tmp_path / "nothing.ckpt"))


def test_evaluate_cli_from_a_tensor_bundle(tmp_path):
    """the reference's evaluation command line end to end: restore from a tensor bundle, 3 votes, outputs on disk"""
    from scanobjectnn_amd import data_utils as DU
    from scanobjectnn_amd.pointnet2 import evaluate_scenennobjects as EV
    from scanobjectnn_amd.pointnet2 import pointnet2_cls_ssg as m
    raw = synth_clouds(24, 640, seed=21) * 3.0 + 1.5            # NOT centred / normalised: the CLI has to do it
    labels = synth_labels(24, seed=21)
    np.savez(tmp_path / "test.npz", data=raw, label=labels)
    x = torch.zeros((2, 512, 3), device=DEV)
    net = Model(m.get_model, device=DEV, seed=5).build(x)
    _randomise(net, 9)
    prefix = str(tmp_path / "model.ckpt")
    _indep_bundle(prefix, _tf_names(net.state_dict(), ema_shadow=False))
    dump = tmp_path / "dump"
    args = EV.parse_args(["--model", "pointnet2_cls_ssg", "--num_point", "512", "--batch_size", "8", "--num_votes", "3",
                          "--model_path", prefix, "--test_file", str(tmp_path / "test.npz"), "--dump_dir", str(dump)])
    ev = EV.evaluate(args)
    assert ev["pred"].shape == (24,) and 0.0 <= ev["accuracy"] <= 1.0 and np.isfinite(ev["mean_loss"]) and ev["mean_loss"] > 0
    lines = open(dump / "pred_label.txt").read().splitlines()
    assert len(lines) == 24 and all(len(l.split(", ")) == 2 and set(l.split(", ")) <= set(EV.SHAPE_NAMES) for l in lines)
    log = open(dump / "log_evaluate.txt").read()
    for key in ("Model restored.", "total seen: 24", "eval mean loss:", "eval accuracy:", "eval avg class acc:", "toilet:"):
        assert key in log, key
    # the same protocol by hand: np.random.seed(0) -> one point subset + one cloud order (get_current_data_h5), data
    # centred and normalised first, logits summed over the 3 rotations
    want = DU.normalize_data(DU.center_data(raw.copy()))
    cur, lab = DU.get_current_data_h5(want, labels, 512, rng=np.random.RandomState(0))
    assert np.array_equal(ev["label"], lab)
    r = EV.eval_one_epoch(net, cur, lab, 8, num_votes=3, device=DEV)
    assert np.array_equal(r["pred"], ev["pred"])
    assert [l.split(", ")[1] for l in lines] == [EV.SHAPE_NAMES[i] for i in lab]


@pytest.mark.parametrize("name", ["pointnet2_cls_bga", "dgcnn_bga"])
def test_evaluate_seg_cli_from_a_tensor_bundle(name, tmp_path):
    """`pointnet2/evaluate_seg_scenennobjects.py:33-53,179-340` end to end for both background-aware models: restore
    from a tensor bundle, 2 votes, class + mask metrics, outputs on disk; the same protocol by hand gives the same
    predictions (file order, first num_point points, masks binarised, logits of both heads summed over the votes)"""
    import importlib
    from scanobjectnn_amd import data_utils as DU
    from scanobjectnn_amd import provider
    from scanobjectnn_amd.pointnet2 import evaluate_seg_scenennobjects as EVS
    from scanobjectnn_amd.pointnet2.train import MODELS
    from scanobjectnn_amd.synth import synth_masks
    mod = importlib.import_module(MODELS[name])
    n_pts = 256 if name.startswith("dgcnn") else 512
    raw = synth_clouds(12, n_pts + 64, seed=31) * 2.0 + 0.7       # NOT centred / normalised: the CLI has to do it
    labels = synth_labels(12, seed=31)
    parts = synth_masks(12, n_pts + 64, seed=31) * 3 - 1         # raw masks: -1 = background, 2 = an object part id
    np.savez(tmp_path / "test.npz", data=raw, label=labels, mask=parts)
    net = Model(mod.get_model, device=DEV, seed=5).build(torch.zeros((2, n_pts, 3), device=DEV))
    _randomise(net, 9)
    prefix = str(tmp_path / "model.ckpt")
    _indep_bundle(prefix, _tf_names(net.state_dict(), ema_shadow=name.startswith("dgcnn")))
    dump = tmp_path / "dump"
    args = EVS.parse_args(["--model", name, "--num_point", str(n_pts), "--batch_size", "4", "--num_votes", "2",
                           "--model_path", prefix, "--test_file", str(tmp_path / "test.npz"), "--dump_dir", str(dump)])
    ev = EVS.evaluate(args)
    assert ev["pred"].shape == (12,) and ev["seg_pred"].shape == (12, n_pts)
    assert 0.0 <= ev["accuracy"] <= 1.0 and 0.0 <= ev["seg_accuracy"] <= 1.0 and np.isfinite(ev["mean_loss"])
    lines = open(dump / "pred_label.txt").read().splitlines()
    assert len(lines) == 12 and [l.split(", ")[1] for l in lines] == [EVS.SHAPE_NAMES[i] for i in labels]
    log = open(dump / "log_evaluate.txt").read()
    for key in ("Model restored.", "total seen: 12", "eval mean loss:", "eval accuracy:", "eval avg class acc:",
                "seg accuracy: %f" % ev["seg_accuracy"], "toilet:"):
        assert key in log, key
    # by hand
    data = DU.normalize_data(DU.center_data(raw.copy()))[:, :n_pts]
    mask = DU.convert_to_binary_mask(parts)[:, :n_pts]
    assert set(np.unique(mask)) <= {0, 1}
    cls_pred, seg_pred = [], []
    with torch.no_grad():
        for lo in range(0, 12, 4):
            pts = torch.from_numpy(data[lo:lo + 4]).to(DEV)
            c = s = 0
            for v in range(2):
                cp, sp = net(provider.rotate_point_cloud_by_angle(pts, v / 2.0 * np.pi * 2).contiguous(), is_training=False)
                c, s = c + cp, s + sp
            cls_pred.append(c.argmax(1).cpu().numpy())
            seg_pred.append(s.argmax(2).cpu().numpy())
    assert np.array_equal(np.concatenate(cls_pred), ev["pred"])
    assert np.array_equal(np.concatenate(seg_pred), ev["seg_pred"])
    assert ev["seg_accuracy"] == (np.concatenate(seg_pred) == mask).sum() / (12.0 * n_pts)
