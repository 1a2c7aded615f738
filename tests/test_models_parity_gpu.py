"""Logit parity of the product models (HIP ops + fused fp32-MFMA MLP) against the CPU restatement
(oracle/ref_models.py) with identical weights: |Δ| <= 1e-4 (BASELINE.json north_star), in eval mode (moving
statistics) and in training mode (batch statistics; dropout disabled on both sides because the two RNGs
cannot be aligned).  The restatement is evaluated in float64 -- the high-precision truth: two fp32 paths with
different summation orders can legitimately sit 2e-4 apart through a dozen batch-normalised layers, so each is
judged against the truth, not against the other.  Gradients of the training loss agree to 5e-3 of the
gradient's max (weight gradients behind a batch norm are heavily cancelling sums).
"""
import os
import statistics

import numpy as np
import pytest
import torch

import decisions as DEC
from oracle import oracle as O
from oracle import ref_models as R
from scanobjectnn_amd.graph import Model
from scanobjectnn_amd.synth import synth_clouds, synth_labels, synth_masks

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4
# Round 4: NO tolerance is derived from the fused path itself any more.  An fp32 evaluation of these
# nets differs from float64 in two ways: (i) rounding of the arithmetic, and (ii) a handful of DISCRETE decisions that sit
# within rounding of a tie (a ReLU at 1e-7, two pool members 5e-7 apart) and fall the other way -- each moves whole
# gradient rows by O(1), a different handful in every path and for every seed.  (ii) is not an error of the arithmetic,
# and instead of being absorbed into a tolerance it is now REMOVED: tests/decisions.py reads back the decisions each
# path took (fused kernels: from the raw layer outputs / BN coefficients / arg-max rows the autograd nodes keep;
# layer-by-layer path: from its relu / amax calls), oracle/ref_models.py evaluates the float64 truth WITH those decisions,
# every decision that differs from the float64 run's own is counted and checked to be a rounding-level tie, and what
# remains -- pure arithmetic error -- is compared between the fused path and the plain fp32 paths.
TIE = 2e-4          # |float64 pre-activation| of a flipped ReLU / gap of a flipped pool member: rounding level (2 x TOL)
GRAD_RESOLUTION = 2e-5   # relative Frobenius error below which two fp32 evaluations of these gradients are not separable
GRAD_CEILING = 3e-5      # hard ceiling on the fused path's gradient error once the decisions agree (measured: 6e-6 ... 1.1e-5)
SEEDS = [int(v) for v in os.environ.get("PCOPS_PARITY_SEEDS", "21,22,23,24,25,26,27,28").split(",")]


def _randomise(net, seed):
    """non-trivial BN affine + moving stats (defaults are 0/1 and would hide bugs).  Biases: a bias in front
    of a batch norm is redundant, initialised to 0 by the reference and receives an analytically zero
    gradient, so it stays ~0 in any real checkpoint -- it gets a tiny value here (a LARGE one would only test
    how fp32 batch norm degrades when |mean| >> std, which no fp32 implementation survives to 1e-4); biases of
    the BN-free output layers get full-size values."""
    g = torch.Generator().manual_seed(seed)
    names = dict(net.named_parameters())
    with torch.no_grad():
        for name, p in list(net.named_parameters()) + list(net.named_buffers()):
            if name.endswith("gamma"):
                p.copy_((0.5 + torch.rand(p.shape, generator=g)).to(p.device))
            elif name.endswith("beta"):
                p.copy_((0.2 * torch.randn(p.shape, generator=g)).to(p.device))
            elif name.endswith("biases"):
                has_bn = name[:-len("biases")] + "bn/gamma" in names
                p.copy_(((0.01 if has_bn else 0.2) * torch.randn(p.shape, generator=g)).to(p.device))
            elif name.endswith("moving_mean") or name.endswith("pop_mean"):
                p.copy_((0.1 * torch.randn(p.shape, generator=g)).to(p.device))
            elif name.endswith("moving_variance") or name.endswith("pop_var"):
                p.copy_((0.5 + torch.rand(p.shape, generator=g)).to(p.device))
            elif name.endswith("transform_XYZ/weights"):
                p.copy_((0.01 * torch.randn(p.shape, generator=g)).to(p.device))


def _record(key, err, floor, **more):
    """keep the measured numbers next to the other GPU artefacts (gpurun_out/ is merged back by gpurun)"""
    import json
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(path, exist_ok=True)
        f = os.path.join(path, "parity_floor.json")
        d = json.load(open(f)) if os.path.exists(f) else {}
        d[key] = dict({"err_fused": err, "fp32_floor": floor}, **more)
        json.dump(d, open(f, "w"), indent=1)
    except OSError:
        pass


def _masked_logit_errors(net, sd, x, c, pick, ref_fn, monkeypatch, fused, D_fused):
    """Training-mode logits of the SAME net through three fp32 paths -- the fused kernels (`fused`, decisions
    `D_fused`), the layer-by-layer product path on the GPU (library GEMM + torch batch norm) and the fp32 CPU
    restatement -- each against the float64 truth evaluated (a) on its own decisions (`err`) and (b) WITH THE
    DECISIONS THAT PATH TOOK (`masked`: what is left is the path's arithmetic), plus the count of decisions that differ
    and the proof that each is a rounding-level tie.  -> {path: {...}}"""
    from scanobjectnn_amd.pointnet2 import tf_util as t2
    rms = lambda t: float(t.double().pow(2).mean().sqrt())   # noqa: E731
    P64 = R.params_from_state_dict(sd, dtype=torch.float64, device=DEV)    # float64 torch algebra on the GPU, C-oracle geometry
    x64 = torch.from_numpy(c).double().to(DEV)
    with torch.no_grad():
        own = pick(ref_fn(x64, P64, True))

    def against_truth(out, D):
        rep = {}
        with torch.no_grad(), R.imposing(D, None, rep):
            forced = pick(ref_fn(x64, P64, True))
        d_own, d_forced = out.to(own.device).double() - own, out.to(own.device).double() - forced
        return dict(err=d_own.abs().max().item(), rms=rms(d_own), masked=d_forced.abs().max().item(),
                    masked_rms=rms(d_forced), flips=DEC.summarise(rep, TIE))

    res = {"fused": against_truth(fused, D_fused)}
    rec = DEC.Recorder(net, DEV)
    monkeypatch.setattr(t2, "FUSED_MLP", False)
    net.load_state_dict(sd)
    with torch.no_grad(), rec.recording():
        plain = pick(net(x, is_training=True, bn_decay=0.9))
    monkeypatch.setattr(t2, "FUSED_MLP", True)
    net.load_state_dict(sd)
    res["gpu_layerwise"] = against_truth(plain, rec.decisions())
    P32 = R.params_from_state_dict(sd, dtype=torch.float32)
    D32 = R.Decisions()
    with torch.no_grad(), R.imposing(None, D32, None):
        cpu = pick(ref_fn(torch.from_numpy(c), P32, True))
    res["cpu_fp32"] = against_truth(cpu, D32)
    return res


def _train_logits_over_seeds(key, build, pick, ref_fn, monkeypatch):
    """Training-mode per-point logits (17 batch-normalised layers deep) over SEEDS: the bar is 1e-4 outright on every seed
    (round 5; rounds 3-4 allowed max(1e-4, the plain-fp32 floor)).  Every path is judged on the float64 truth evaluated
    with ITS OWN decisions, so that a flipped ReLU in one of them is not mistaken for arithmetic; the floor is what plain
    fp32 evaluations of these nets show: the layer-by-layer product path (library GEMM + torch batch norm) and the fp32
    CPU restatement, over all seeds.  The maximum over ~3e4 outputs moves +-15 % from one fp32 evaluation to the next,
    which is why the floor is the plain paths' worst over the seeds; the (stable) RMS error is held to the plain paths'
    on average over the seeds.  (Round 4 made this bar reachable: tools/diag_stage_noise.py located the fused path's
    +8 % RMS in the small-row GEMM's single K-long accumulation chain -- the plain path's small GEMM splits K four
    ways -- and csrc/mlp.hip gemm_rt_kernel now sums K in 32-wide blocks: stage noise 1.08 -> 1.00 of plain.)"""
    cases = []
    for seed in SEEDS:
        net, x, c = build(seed)
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        rec = DEC.Recorder(net, DEV)
        with torch.enable_grad(), rec.recording():   # grad mode: the fused nodes only keep what a backward needs
            out = pick(net(x, is_training=True, bn_decay=0.9)).detach()
        res = _masked_logit_errors(net, sd, x, c, pick, ref_fn, monkeypatch, out, rec.decisions())
        res["seed"] = seed
        cases.append(res)
    plain = ("gpu_layerwise", "cpu_fp32")
    floor = max(r[p]["masked"] for r in cases for p in plain)
    _record(key, max(r["fused"]["err"] for r in cases), floor, seeds=cases,
            worst_fused_masked=max(r["fused"]["masked"] for r in cases),
            mean_rms={p: sum(r[p]["masked_rms"] for r in cases) / len(cases) for p in ("fused",) + plain})
    for r in cases:
        for path in ("fused",) + plain:
            assert r[path]["flips"]["all_ties"], (r["seed"], path, r[path]["flips"])   # every flip is a rounding-level tie
        # 1e-4 OUTRIGHT on every seed (VERDICT r4 #5i): against the truth on the path's own decisions AND, flips included,
        # against the truth's own -- the plain-fp32 floor (1.97e-4 on the worst seed) is recorded, it is no longer the bar
        assert r["fused"]["masked"] <= TOL, (r["seed"], r["fused"], floor)
        assert r["fused"]["err"] <= TOL, r
    # the RMS error (stable to a few per cent per seed, unlike the maximum): on average over the seeds the fused path is
    # no noisier than the noisier of the two plain fp32 evaluations -- no factor
    mean_rms = {p: sum(r[p]["masked_rms"] for r in cases) / len(cases) for p in ("fused",) + plain}
    assert mean_rms["fused"] <= max(mean_rms[p] for p in plain), mean_rms


def _no_dropout(monkeypatch):
    from scanobjectnn_amd.pointnet2 import tf_util as t2
    from scanobjectnn_amd.dgcnn import tf_util as td
    ident = lambda inputs, is_training, scope, keep_prob=0.5, noise_shape=None: inputs  # noqa: E731
    monkeypatch.setattr(t2, "dropout", ident)
    monkeypatch.setattr(td, "dropout", ident)


@pytest.mark.parametrize("training", [False, True])
@pytest.mark.parametrize("name", ["ssg", "msg"])
def test_pointnet2_cls_logits(name, training, monkeypatch):
    from scanobjectnn_amd.pointnet2 import pointnet2_cls_msg, pointnet2_cls_ssg
    mod, ref = {"ssg": (pointnet2_cls_ssg, R.pointnet2_cls_ssg), "msg": (pointnet2_cls_msg, R.pointnet2_cls_msg)}[name]
    _no_dropout(monkeypatch)
    c = synth_clouds(16, 1024, seed=3)   # >=16 so the FC-level batch norm is not degenerate
    x = torch.from_numpy(c).to(DEV)
    net = Model(mod.get_model, device=DEV, seed=1).build(x)
    _randomise(net, 5)
    P = R.params_from_state_dict(net.state_dict(), dtype=torch.float64)
    with torch.no_grad():
        logits, _ = net(x, is_training=training, bn_decay=0.9)
        want = ref(torch.from_numpy(c).double(), P, training)
    assert (logits.cpu().double() - want).abs().max().item() <= TOL


def test_pointnet2_bga_logits_and_mask_eval(monkeypatch):
    """eval mode (what evaluate_*.py runs): class logits and mask logits hold the 1e-4 bar outright"""
    from scanobjectnn_amd.pointnet2 import pointnet2_cls_bga as m
    _no_dropout(monkeypatch)
    c = synth_clouds(16, 1024, seed=4)
    x = torch.from_numpy(c).to(DEV)
    net = Model(m.get_model, device=DEV, seed=2).build(x)
    _randomise(net, 6)
    P = R.params_from_state_dict(net.state_dict(), dtype=torch.float64)
    with torch.no_grad():
        cls, seg = net(x, is_training=False, bn_decay=0.9)
        wc, ws = R.pointnet2_cls_bga(torch.from_numpy(c).double(), P, False)
    assert (cls.cpu().double() - wc).abs().max().item() <= TOL
    assert (seg.cpu().double() - ws).abs().max().item() <= TOL


def test_pointnet2_bga_logits_and_mask_train(monkeypatch):
    """batch statistics: class logits <= 1e-4 outright; the mask logits against max(1e-4, plain-fp32 floor)"""
    from scanobjectnn_amd.pointnet2 import pointnet2_cls_bga as m
    _no_dropout(monkeypatch)

    def build(seed):
        c = synth_clouds(16, 1024, seed=seed)
        x = torch.from_numpy(c).to(DEV)
        net = Model(m.get_model, device=DEV, seed=seed - 19).build(x)
        _randomise(net, seed - 15)
        sd0 = {k: v.clone() for k, v in net.state_dict().items()}
        P = R.params_from_state_dict(sd0, dtype=torch.float64, device=DEV)
        with torch.no_grad():
            cls = net(x, is_training=True, bn_decay=0.9)[0]
            wc = R.pointnet2_cls_bga(torch.from_numpy(c).double().to(DEV), P, True)[0]
        assert (cls.double() - wc).abs().max().item() <= TOL, seed
        net.load_state_dict(sd0)              # the training-mode pass moved the BN moving statistics
        return net, x, c
    _train_logits_over_seeds("bga_mask_train", build, lambda o: o[1], R.pointnet2_cls_bga, monkeypatch)


def test_pointnet2_partseg_logits_eval(monkeypatch):
    """pointnet2_cls_partseg: the SA + FP stacks with the global feature propagated from the single l3
    point; per-point part logits against the float64 restatement, eval mode"""
    from scanobjectnn_amd.pointnet2 import pointnet2_cls_partseg as m
    _no_dropout(monkeypatch)
    c = synth_clouds(16, 1024, seed=9)
    x = torch.from_numpy(c).to(DEV)
    net = Model(m.get_model, device=DEV, seed=4).build(x)
    _randomise(net, 10)
    P = R.params_from_state_dict(net.state_dict(), dtype=torch.float64)
    with torch.no_grad():
        seg = net(x, is_training=False, bn_decay=0.9)
        want = R.pointnet2_cls_partseg(torch.from_numpy(c).double(), P, False)
    assert seg.shape == (16, 1024, 6)
    assert (seg.cpu().double() - want).abs().max().item() <= TOL


def test_pointnet2_partseg_logits_train(monkeypatch):
    from scanobjectnn_amd.pointnet2 import pointnet2_cls_partseg as m
    _no_dropout(monkeypatch)

    def build(seed):
        c = synth_clouds(16, 1024, seed=seed)
        x = torch.from_numpy(c).to(DEV)
        net = Model(m.get_model, device=DEV, seed=seed - 17).build(x)
        _randomise(net, seed - 11)
        return net, x, c
    _train_logits_over_seeds("partseg_train", build, lambda o: o, R.pointnet2_cls_partseg, monkeypatch)


@pytest.mark.parametrize("training", [False, True])
@pytest.mark.parametrize("name", ["dgcnn", "dgcnn_bga"])
def test_dgcnn_logits(name, training, monkeypatch):
    from scanobjectnn_amd.dgcnn import dgcnn, dgcnn_bga
    from scanobjectnn_amd.dgcnn import tf_util as td
    _no_dropout(monkeypatch)
    mod = {"dgcnn": dgcnn, "dgcnn_bga": dgcnn_bga}[name]
    c = synth_clouds(12, 256, seed=5)
    x = torch.from_numpy(c).to(DEV)
    net = Model(mod.get_model, device=DEV, seed=4).build(x)
    _randomise(net, 9)
    P = R.params_from_state_dict(net.state_dict(), dtype=torch.float64)
    graphs = []
    real = td.knn_graph

    def recording(point_cloud, k=20, seed=None):
        nn = real(point_cloud, k=k, seed=seed)
        # the HIP graph (seeded by the previous layer's, dgcnn.py) is bit-exact w.r.t. the oracle GIVEN the same input tensor
        inp = point_cloud.detach().reshape(point_cloud.shape[0], point_cloud.shape[1], -1).cpu().numpy()
        np.testing.assert_array_equal(nn.cpu().numpy(), O.knn_graph(inp, k))
        graphs.append(nn.cpu().numpy())
        return nn

    monkeypatch.setattr(td, "knn_graph", recording)
    with torch.no_grad():
        out = net(x, is_training=training, bn_decay=0.9)
    assert len(graphs) == 5
    if name == "dgcnn":
        want = R.dgcnn(torch.from_numpy(c).double(), P, training, nn_list=graphs)
        assert (out[0].cpu().double() - want).abs().max().item() <= TOL
    else:
        wc, ws = R.dgcnn_bga(torch.from_numpy(c).double(), P, training, nn_list=graphs)
        assert (out[0].cpu().double() - wc).abs().max().item() <= TOL
        assert (out[1].cpu().double() - ws).abs().max().item() <= TOL


# ---------------------------------------------------------------------------------------------------------------
# model-level GRADIENT parity for every in-scope model (training mode, batch statistics, dropout off)
def _grad_errors(got, want):
    """global relative Frobenius error of a {name: gradient} dict against the float64 restatement's, and the worst
    tensor.  Biases in front of a batch norm (analytically zero gradient) are left out."""
    num = den = 0.0
    worst = ("", 0.0)
    for name, g in got.items():
        ref = want[name]
        if name.endswith("biases") and name[:-len("biases")] + "bn/gamma" in got:
            continue
        e = (g.to(ref.device).double() - ref).norm().item()
        r = ref.norm().item()
        if r < 1e-9:
            assert e < 1e-5, (name, e, r)
            continue
        if e / r > worst[1]:
            worst = (name, e / r)
        num += e * e
        den += r * r
    return (num / den) ** 0.5, worst


GRAD_MODELS = ["ssg", "bga", "msg", "dgcnn", "dgcnn_bga"]


def _grad_case(name, seed, monkeypatch, batch=None, num_point=None, paths=("fused", "layer")):
    """one (model, seed): gradients of the training loss through the fused kernels and through the layer-by-layer path,
    each against the float64 truth on its own decisions (`e_*`) and against the float64 truth evaluated WITH THE
    DECISIONS THAT PATH TOOK (`em_*`), and what those decisions were"""
    from scanobjectnn_amd.dgcnn import dgcnn, dgcnn_bga
    from scanobjectnn_amd.dgcnn import tf_util as td
    from scanobjectnn_amd.pointnet2 import pointnet2_cls_bga, pointnet2_cls_msg, pointnet2_cls_ssg
    from scanobjectnn_amd.pointnet2 import tf_util as t2
    mod, ref, n_pts, has_mask = {
        "ssg": (pointnet2_cls_ssg, R.pointnet2_cls_ssg, 1024, False),
        "bga": (pointnet2_cls_bga, R.pointnet2_cls_bga, 1024, True),
        "msg": (pointnet2_cls_msg, R.pointnet2_cls_msg, 1024, False),
        "dgcnn": (dgcnn, R.dgcnn, 256, False),
        "dgcnn_bga": (dgcnn_bga, R.dgcnn_bga, 256, True)}[name]
    B = batch or 16
    n_pts = num_point or n_pts
    c = synth_clouds(B, n_pts, seed=seed)
    y = torch.from_numpy(synth_labels(B, seed=seed))
    mask = torch.from_numpy(synth_masks(B, n_pts, seed=seed)) if has_mask else None
    x = torch.from_numpy(c).to(DEV)
    net = Model(mod.get_model, device=DEV, seed=seed - 15).build(x)
    _randomise(net, seed - 9)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    rec = DEC.Recorder(net, DEV)
    is_dg = name.startswith("dgcnn")
    graphs = []
    real = td.knn_graph

    def recording(point_cloud, k=20, seed=None):
        nn = real(point_cloud, k=k, seed=seed)
        graphs.append(nn.cpu().numpy())
        return nn

    def product_grads():
        net.load_state_dict(sd)
        net.zero_grad(set_to_none=True)
        with rec.recording():
            out = net(x, is_training=True, bn_decay=0.9)
        loss = (mod.get_loss(out[0], out[1], y.to(DEV), mask.to(DEV))[0] if has_mask
                else mod.get_loss(out[0], y.to(DEV)))
        loss.backward()
        return (loss.item(), rec.decisions(),
                {k[len("graph."):]: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None})

    if is_dg:
        monkeypatch.setattr(td, "knn_graph", recording)
    loss_fused, D_fused, g_fused = product_grads()
    graphs_fused = list(graphs)
    # the layer-by-layer path (library GEMM + torch batch norm): the fp32 yardstick.  DGCNN: on the neighbour graphs of the
    # fused run (which the truth is evaluated with too -- a 20th-neighbour tie that falls the other way is a different
    # network; the graphs themselves are bit-exact against the oracle, test_dgcnn_logits / test_knn_gpu.py)
    D_layer = g_layer = None
    if "layer" in paths:
        monkeypatch.setattr(t2, "FUSED_MLP", False)
        if is_dg:
            replay = iter([torch.from_numpy(g).to(DEV) for g in graphs_fused])
            monkeypatch.setattr(td, "knn_graph", lambda point_cloud, k=20, seed=None: next(replay))
        _, D_layer, g_layer = product_grads()
        monkeypatch.setattr(t2, "FUSED_MLP", True)
    monkeypatch.setattr(td, "knn_graph", real)

    kw = {"nn_list": graphs_fused} if is_dg else {}

    def truth(D=None, record=None, report=None):
        # float64 autograd of the restatement: torch algebra on the GPU, geometry from the C oracle
        P = {k: v.requires_grad_(v.is_floating_point())
             for k, v in R.params_from_state_dict(sd, dtype=torch.float64, device=DEV).items()}
        with R.imposing(D, record, report):
            want = ref(torch.from_numpy(c).double().to(DEV), P, True, **kw)
        loss = (mod.get_loss(want[0], want[1], y.to(DEV), mask.to(DEV))[0] if has_mask else mod.get_loss(want, y.to(DEV)))
        loss.backward()
        return loss.item(), {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in P.items()
                             if v.is_floating_point() and k in g_fused}

    D64 = R.Decisions()
    loss_ref, g64 = truth(record=D64)
    assert abs(loss_fused - loss_ref) <= 1e-4
    out = {"seed": seed, "loss_fused": loss_fused, "loss_ref": loss_ref}
    if is_dg:
        # the third kind of discrete decision: the neighbour graphs.  The truth above is evaluated ON the product's graphs
        # (each bit-exact against the oracle given its input tensor: test_dgcnn_logits, test_knn_gpu.py); here they are
        # compared with the graphs the float64 run builds on ITS OWN features -- rows (cloud, point) whose neighbour SET
        # differs are counted per graph: a 20th-neighbour near-tie decided the other way by a 1e-7 feature difference
        own_graphs = []
        real_knn = O.knn_graph

        def rec_knn(xx, k):
            g = real_knn(xx, k)
            own_graphs.append(g)
            return g
        monkeypatch.setattr(O, "knn_graph", rec_knn)
        with torch.no_grad():
            ref(torch.from_numpy(c).double().to(DEV), R.params_from_state_dict(sd, dtype=torch.float64, device=DEV), True)
        monkeypatch.setattr(O, "knn_graph", real_knn)
        assert len(own_graphs) == len(graphs_fused) == 5
        out["graph_rows_differing"] = [int((np.sort(a, -1) != np.sort(b, -1)).any(-1).sum())
                                       for a, b in zip(own_graphs, graphs_fused)]
        out["graph_rows"] = int(graphs_fused[0].shape[0] * graphs_fused[0].shape[1])
        assert sum(out["graph_rows_differing"]) <= 0.01 * 5 * out["graph_rows"], out       # few, and only near-ties can differ
    for path, D, g in (("fused", D_fused, g_fused), ("layer", D_layer, g_layer)):
        if path not in paths:
            continue
        # the read-back covers exactly the layers of the network, each with the kind of decision it has
        assert set(D.relu) == set(D64.relu) and set(D.pool) == set(D64.pool), \
            (path, sorted(set(D.relu) ^ set(D64.relu)), sorted(set(D.pool) ^ set(D64.pool)))
        rep = {}
        _, g64_forced = truth(D, report=rep)
        e, worst = _grad_errors(g, g64)
        em, worst_m = _grad_errors(g, g64_forced)
        if batch:       # bench-size runs: the per-variable picture too (relative error of every variable's gradient)
            out["per_variable_" + path] = {k: ((g[k].double() - g64_forced[k]).norm() / g64_forced[k].norm().clamp_min(1e-30)).item()
                                           for k in g64_forced}
            out["grad_norm_" + path] = {k: g[k].double().norm().item() for k in g64_forced}
        out.update({"e_" + path: e, "em_" + path: em, "worst_" + path: worst, "worst_masked_" + path: worst_m,
                    "flips_" + path: DEC.summarise(rep, TIE)})
    return out


@pytest.mark.parametrize("name", GRAD_MODELS)
def test_model_training_gradients(name, monkeypatch):
    """d(loss)/d(every variable) of pointnet2_cls_ssg / _bga / _msg / dgcnn / dgcnn_bga against float64 autograd of
    the restatement, over len(SEEDS) >= 8 seeds (weights, BN state, clouds, labels all re-drawn).  Per seed:
      * the decisions each path took are read back; those that differ from the float64 run's are COUNTED and each must be
        a rounding-level tie (float64 pre-activation / pool gap <= TIE);
      * with the float64 truth evaluated on the path's own decisions, the fused kernels' gradient error -- now purely
        arithmetic -- must be within 1.5x the layer-by-layer path's (library GEMM + torch batch norm) or at the fp32
        resolution of these sums;
    and over the seeds the median UNMASKED error of the fused path must be <= 2x the layer-by-layer path's (flips hit
    both paths alike).
    No bar involves a second evaluation of the fused path (round 3's `fused_realisations` is gone)."""
    _no_dropout(monkeypatch)
    cases = [_grad_case(name, seed, monkeypatch) for seed in SEEDS]
    import json
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(path, exist_ok=True)
        f = os.path.join(path, "parity_flips.json")
        d = json.load(open(f)) if os.path.exists(f) else {}
        d["grad_" + name] = {"tie": TIE, "seeds": cases,
                             "median_unmasked_ratio": statistics.median(c["e_fused"] / c["e_layer"] for c in cases),
                             "ratio_of_median_unmasked_errors": statistics.median(c["e_fused"] for c in cases)
                             / statistics.median(c["e_layer"] for c in cases),
                             "max_masked_ratio": max(c["em_fused"] / c["em_layer"] for c in cases)}
        json.dump(d, open(f, "w"), indent=1)
    except OSError:
        pass
    for c in cases:
        for path in ("fused", "layer"):
            f = c["flips_" + path]
            assert f["all_ties"], (path, c)                      # every decision that differs is a rounding-level tie ...
            assert f["relu_flips"] <= max(8, 2e-5 * f["relu_elements"]), (path, c)     # ... and there are few of them
            assert f["pool_flips"] + f["active_flips"] <= max(8, 2e-5 * f["pool_elements"]), (path, c)
        # the arithmetic: error against the truth on the SAME decisions -- an absolute ceiling at fp32 resolution of these
        # sums (no bar without one) and no worse than the layer-by-layer path
        assert c["em_fused"] <= GRAD_CEILING, c
        assert c["em_fused"] <= max(1.5 * c["em_layer"], GRAD_RESOLUTION), c
        # flips included: an UNMASKED ceiling too (a defect confined to the elements classified as flips must not
        # hide behind the mask) -- 2e-2, the bar of rounds 1-2 (measured worst 1.5e-2, dgcnn, one head-level near-tie), and
        # every masked flip is a float64 near-tie on both paths (all_ties above)
        assert c["e_fused"] <= 2e-2, c
    # flips included, the two paths are hit alike.  Per seed the ratio e_fused / e_layer is a coin toss between << 1 and
    # >> 1 (whichever path met the nastier tie: 0.002 ... 127 in the committed record, both directions), so the MEDIAN
    # ERRORS of the two paths are compared, not the median of that ratio (recorded as median_unmasked_ratio)
    assert statistics.median(c["e_fused"] for c in cases) <= 2.0 * statistics.median(c["e_layer"] for c in cases), cases


# ---------------------------------------------------------------------------------------------------------------
# pointnet_sa_module branches no model in scope takes (`pointnet2/utils/pointnet_util.py:128-151`): pooling =
# avg | weighted_avg | max_and_avg, mlp2, knn=True -- against a float64 torch restatement of the reference lines
def _sa_reference(xyz, points, npoint, radius, nsample, mlp, mlp2, P, training, pooling, knn):
    """float64 restatement of pointnet_sa_module (:87-154) on the C-oracle geometry"""
    fps = O.farthest_point_sample(npoint, xyz.float().numpy())
    new_xyz = R.batch_gather(xyz, R._idx(fps))
    if knn:
        _, idx = O.knn_point(nsample, xyz.float().numpy(), new_xyz.float().numpy())
    else:
        idx, _ = O.query_ball_point(radius, nsample, xyz.float().numpy(), new_xyz.float().numpy())
    idx = R._idx(idx)
    grouped_xyz = R.batch_gather(xyz, idx) - new_xyz.unsqueeze(2)
    new_points = grouped_xyz if points is None else torch.cat([grouped_xyz, R.batch_gather(points, idx)], -1)
    for i in range(len(mlp)):
        new_points = R.dense(new_points, P, "sa/conv%d" % i, training)
    if pooling == "max":
        new_points = new_points.amax(dim=2, keepdim=True)
    elif pooling == "avg":
        new_points = new_points.mean(dim=2, keepdim=True)
    elif pooling == "weighted_avg":
        dists = grouped_xyz.norm(dim=-1, p=2, keepdim=True)
        e = torch.exp(-dists * 5)
        new_points = (new_points * (e / e.sum(dim=2, keepdim=True))).sum(dim=2, keepdim=True)
    elif pooling == "max_and_avg":
        new_points = torch.cat([new_points.mean(dim=2, keepdim=True), new_points.amax(dim=2, keepdim=True)], -1)
    if mlp2 is not None:
        for i in range(len(mlp2)):
            new_points = R.dense(new_points, P, "sa/conv_post_%d" % i, training)
    return new_xyz, new_points.squeeze(2), idx


@pytest.mark.parametrize("training", [False, True])
@pytest.mark.parametrize("pooling,mlp2,knn", [("avg", None, False), ("weighted_avg", None, False),
                                              ("max_and_avg", None, False), ("max", [64, 32], False),
                                              ("max", None, True), ("avg", [32], True)])
def test_sa_module_branches(pooling, mlp2, knn, training):
    from scanobjectnn_amd.pointnet2.pointnet_util import pointnet_sa_module
    c = synth_clouds(6, 512, seed=31)
    feats = np.random.default_rng(2).standard_normal((6, 512, 16)).astype(np.float32)
    x, f = torch.from_numpy(c).to(DEV), torch.from_numpy(feats).to(DEV)

    def get_model(point_cloud, is_training, bn_decay=None):
        return pointnet_sa_module(point_cloud, f, npoint=64, radius=0.3, nsample=16, mlp=[32, 64], mlp2=mlp2,
                                  group_all=False, is_training=is_training, bn_decay=bn_decay, scope="sa",
                                  pooling=pooling, knn=knn)
    net = Model(get_model, device=DEV, seed=3).build(x)
    _randomise(net, 14)
    P = R.params_from_state_dict(net.state_dict(), dtype=torch.float64)
    with torch.no_grad():
        new_xyz, new_points, idx = net(x, is_training=training, bn_decay=0.9)
        w_xyz, w_points, w_idx = _sa_reference(torch.from_numpy(c).double(), torch.from_numpy(feats).double(), 64, 0.3,
                                               16, [32, 64], mlp2, P, training, pooling, knn)
    cout = (mlp2[-1] if mlp2 else 64) * (2 if (pooling == "max_and_avg" and not mlp2) else 1)
    assert new_points.shape == (6, 64, cout)
    np.testing.assert_array_equal(idx.cpu().numpy(), w_idx.numpy())                    # integer outputs: bit-exact
    assert (new_xyz.cpu().double() - w_xyz).abs().max().item() == 0.0
    assert (new_points.cpu().double() - w_points).abs().max().item() <= TOL


def test_sa_module_keeps_the_coordinate_gradient():
    """when xyz carries gradient (a T-Net in front, saliency, adversarial perturbation) the SA module must
    differentiate w.r.t. the coordinates like the reference does through GroupPoint / GatherPoint and the centring
    subtraction (`tf_grouping.py:43-47`, `tf_sampling.py:44-48`) -- the fused path has no such gradient and must
    not be taken silently"""
    from scanobjectnn_amd.pointnet2.pointnet_util import pointnet_sa_module
    c = synth_clouds(4, 256, seed=41)
    x = torch.from_numpy(c).to(DEV)

    def get_model(point_cloud, is_training, bn_decay=None):
        return pointnet_sa_module(point_cloud, None, npoint=32, radius=0.4, nsample=16, mlp=[32, 64], mlp2=None,
                                  group_all=False, is_training=is_training, bn_decay=bn_decay, scope="sa")
    net = Model(get_model, device=DEV, seed=5).build(x)
    _randomise(net, 15)
    xg = x.clone().requires_grad_(True)
    _, feats, idx = net(xg, is_training=True, bn_decay=0.9)
    w = torch.randn(feats.shape, generator=torch.Generator().manual_seed(1)).to(DEV)
    (feats * w).sum().backward()
    assert xg.grad is not None and torch.isfinite(xg.grad).all() and xg.grad.abs().max().item() > 0
    # float64 truth with the same (non-differentiable) indices
    P = R.params_from_state_dict(net.state_dict(), dtype=torch.float64)
    xd = torch.from_numpy(c).double().requires_grad_(True)
    fps = R._idx(O.farthest_point_sample(32, c))
    new_xyz = R.batch_gather(xd, fps)
    ii = R._idx(idx.cpu().numpy())
    g = R.batch_gather(xd, ii) - new_xyz.unsqueeze(2)
    for i in range(2):
        g = R.dense(g, P, "sa/conv%d" % i, True)
    (g.amax(dim=2) * w.cpu().double()).sum().backward()
    scale = xd.grad.abs().max().item()
    assert (xg.grad.cpu().double() - xd.grad).abs().max().item() <= 2e-3 * scale
    # same parameters, no coordinate gradient requested: the fused path is allowed again and agrees in value
    with torch.no_grad():
        _, feats2, _ = net(x, is_training=True, bn_decay=0.9)
    assert (feats2 - feats.detach()).abs().max().item() <= 1e-4
