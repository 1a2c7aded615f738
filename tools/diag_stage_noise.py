"""Where along pointnet2_cls_bga (training-mode batch norm) does an fp32 path pick up its distance to float64?
Stage outputs (l1 / l2 / l3 features, class vector, the three FP stacks, mask logits) of the fused path and of the
layer-by-layer path against the float64 restatement, RMS error relative to the stage's RMS value, several seeds.
    python tools/diag_stage_noise.py [seeds...]   (GPU)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ref_models as R  # noqa: E402
from scanobjectnn_amd.graph import Model  # noqa: E402
from scanobjectnn_amd.pointnet2 import pointnet2_cls_bga as M  # noqa: E402
from scanobjectnn_amd.pointnet2 import tf_util as t2  # noqa: E402
from scanobjectnn_amd.synth import synth_clouds  # noqa: E402
from test_models_parity_gpu import _randomise  # noqa: E402

DEV = "cuda:0"


VARIANTS = {          # name -> (FUSED_MLP, {module attribute: value})
    "fused": (True, {}),
    "plain": (False, {}),
    "nopivot": (True, {"fused_mlp.STAT_PIVOT": False}),
    "nocompact": (True, {"fused_mlp.COMPACT_MIN_S": 0}),
    "nofusepool": (True, {"fused_mlp.FUSE_POOL_ROWS": False}),
    "libfc": (True, {"t2.FC_PCOPS": False}),
}


def product_stages(net, x, variant):
    from scanobjectnn_amd import fused_mlp
    fused, toggles = VARIANTS[variant]
    mods = {"fused_mlp": fused_mlp, "t2": t2}
    saved = {}
    for key, val in toggles.items():
        m, a = key.split(".")
        saved[key] = getattr(mods[m], a)
        setattr(mods[m], a, val)
    try:
        return _product_stages(net, x, fused)
    finally:
        for key, val in saved.items():
            m, a = key.split(".")
            setattr(mods[m], a, val)


def _product_stages(net, x, fused):
    stages = {}
    sa, fp, fc, c1 = M.pointnet_sa_module, M.pointnet_fp_module, t2.fully_connected, t2.conv1d

    def w_sa(*a, **k):
        out = sa(*a, **k)
        stages[k["scope"]] = out[1].detach()
        return out

    def w_fp(*a, **k):
        out = fp(*a, **k)
        stages[k["scope"]] = out.detach()
        return out

    def w_fc(*a, **k):
        out = fc(*a, **k)
        stages[k["scope"]] = out.detach()
        return out

    def w_c1(*a, **k):
        out = c1(*a, **k)
        stages[k["scope"]] = out.detach()
        return out
    M.pointnet_sa_module, M.pointnet_fp_module, t2.fully_connected, t2.conv1d = w_sa, w_fp, w_fc, w_c1
    keep = t2.FUSED_MLP
    t2.FUSED_MLP = fused
    t2.dropout_keep = t2.dropout
    t2.dropout = lambda inputs, is_training, scope, keep_prob=0.5, noise_shape=None: inputs
    try:
        with torch.no_grad():
            net(x, is_training=True, bn_decay=0.9)
    finally:
        M.pointnet_sa_module, M.pointnet_fp_module, t2.fully_connected, t2.conv1d = sa, fp, fc, c1
        t2.FUSED_MLP = keep
        t2.dropout = t2.dropout_keep
    return stages


def truth_stages(c, P):
    stages = {}
    sa, fp, dn = R.sa_module, R.fp_module, R.dense

    def w_sa(xyz, points, npoint, radius, nsample, mlp, P_, scope, training, group_all=False):
        out = sa(xyz, points, npoint, radius, nsample, mlp, P_, scope, training, group_all)
        stages[scope] = out[1]
        return out

    def w_fp(xyz1, xyz2, p1, p2, mlp, P_, scope, training):
        out = fp(xyz1, xyz2, p1, p2, mlp, P_, scope, training)
        stages[scope] = out
        return out

    def w_dn(x, P_, scope, training, **k):
        out = dn(x, P_, scope, training, **k)
        if scope in ("fc1", "fc2", "fc3", "seg_fc1", "seg_fc2"):
            stages[scope] = out
        return out
    R.sa_module, R.fp_module, R.dense = w_sa, w_fp, w_dn
    try:
        with torch.no_grad():
            R.pointnet2_cls_bga(torch.from_numpy(c).double().to(DEV), P, True)
    finally:
        R.sa_module, R.fp_module, R.dense = sa, fp, dn
    return stages


def main(seeds):
    rows = {}
    for seed in seeds:
        c = synth_clouds(16, 1024, seed=seed)
        x = torch.from_numpy(c).to(DEV)
        net = Model(M.get_model, device=DEV, seed=seed + 1).build(x)
        _randomise(net, seed + 2)
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        P = R.params_from_state_dict(sd, dtype=torch.float64, device=DEV)
        want = truth_stages(c, P)
        res = {}
        for variant in VARIANTS:
            net.load_state_dict(sd)
            got = product_stages(net, x, variant)
            for k, v in want.items():
                if k in got:
                    d = got[k].double().reshape(v.shape) - v
                    res.setdefault(k, {})[variant] = float(d.pow(2).mean().sqrt() / v.pow(2).mean().sqrt())
        rows[seed] = res
    order = ["layer1", "layer2", "layer3", "fc1", "fc2", "fc3", "fa_layer1", "fa_layer2", "fa_layer3", "seg_fc1", "seg_fc2"]
    names = list(VARIANTS)
    print("relative RMS error per stage, mean over seeds %s; ratio to plain in brackets" % (seeds,))
    print("%-10s" % "stage" + "".join("%22s" % n for n in names))
    for k in order:
        line = "%-10s" % k
        plain = sum(rows[s][k]["plain"] for s in seeds) / len(seeds)
        for n in names:
            v = sum(rows[s][k][n] for s in seeds) / len(seeds)
            line += "%14.3e (%4.2f)" % (v, v / plain)
        print(line)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "stage_noise.json"), "w"), indent=1)


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [4, 5, 6])
