"""Training loop -- restates the step semantics of `pointnet2/train.py:136-171,218-261` (`train_seg.py` for the BGA
models, `train_partseg.py` for part segmentation) with the reference's flag names (`train.py:25-46`), data-parallel
over the GPUs of one node.

  python -m scanobjectnn_amd.pointnet2.train --model pointnet2_cls_ssg --num_point 2048 --batch_size 256 \
         --max_epoch 1 [--train_file x.npz --test_file y.npz]          (synthetic clouds when no file is given)
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m scanobjectnn_amd.pointnet2.train ...

Input pipeline, all on the device: the loaded set is uploaded ONCE, centred and normalised there
(`train.py:100-106` does it on the host: `data_utils.center_data`, `normalize_data`), the per-epoch point subset /
cloud order (`data_utils.get_current_data_h5`) is an index gather on the resident tensor, and rotate + jitter
(`provider.py:34-52,189-200`) run per batch on the device.
Per step: forward (training BN, dropout) -> loss -> backward -> ONE all-reduce of the flat gradient bucket ->
Adam / momentum with the staircase lr; bn_decay follows the reference schedule.  Loss and accuracy are accumulated
on the device and read ONCE per epoch (no per-step host synchronisation).  A checkpoint (`model.pt`, variables under
the reference's TF scope names) is written every epoch after the BN moving statistics were made rank 0's.
"""
import argparse
import importlib
import json
import os
import time

import numpy as np
import torch

from .. import data_utils, dist as D, provider
from .. import train_util as TU
from ..graph import Model
from ..synth import synth_clouds, synth_labels, synth_masks
from . import evaluate_scenennobjects as EV

MODELS = {"pointnet2_cls_ssg": "scanobjectnn_amd.pointnet2.pointnet2_cls_ssg",
          "pointnet2_cls_bga": "scanobjectnn_amd.pointnet2.pointnet2_cls_bga",
          "pointnet2_cls_msg": "scanobjectnn_amd.pointnet2.pointnet2_cls_msg",
          "pointnet2_cls_partseg": "scanobjectnn_amd.pointnet2.pointnet2_cls_partseg",
          "dgcnn": "scanobjectnn_amd.dgcnn.dgcnn", "dgcnn_bga": "scanobjectnn_amd.dgcnn.dgcnn_bga"}


def _flag(v):
    """the reference declares its switches as `default = True` without a type, so ANY string on the command line
    is truthy there; typed properly here"""
    if isinstance(v, bool):
        return v
    if v.lower() in ("1", "true", "yes", "y", "t"):
        return True
    if v.lower() in ("0", "false", "no", "n", "f"):
        return False
    raise argparse.ArgumentTypeError("expected a boolean, got %r" % (v,))


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpu", type=int, default=0, help="GPU to use when not launched by torchrun (train.py:26)")
    p.add_argument("--model", default="pointnet2_cls_ssg", choices=sorted(MODELS))
    p.add_argument("--log_dir", default="log")
    p.add_argument("--with_bg", type=_flag, default=True, help="keep background points of raw .bin objects (:30)")
    p.add_argument("--norm", type=_flag, default=True, help="divide every cloud by its max L2 norm (:31)")
    p.add_argument("--center_data", type=_flag, default=True, help="subtract every cloud's centroid (:32)")
    p.add_argument("--num_class", type=int, default=15)             # :33
    p.add_argument("--num_point", type=int, default=1024)           # :38
    p.add_argument("--max_epoch", type=int, default=250)            # :39
    p.add_argument("--batch_size", type=int, default=16, help="GLOBAL batch, split over the ranks")
    p.add_argument("--learning_rate", type=float, default=0.001)
    p.add_argument("--momentum", type=float, default=0.9)           # :41
    p.add_argument("--optimizer", default="adam", choices=["adam", "momentum"])   # :42
    p.add_argument("--decay_step", type=int, default=200000)
    p.add_argument("--decay_rate", type=float, default=0.7)
    p.add_argument("--normal", action="store_true", help="accepted for CLI compatibility (:46); the reference never reads it")
    p.add_argument("--seg_weight", type=float, default=0.5)         # train_seg.py:35 (typed properly)
    p.add_argument("--train_file", default="")
    p.add_argument("--test_file", default="")
    p.add_argument("--data_path", default="", help="root of the raw objects_bin/ files for pickled split lists")
    p.add_argument("--synthetic_clouds", type=int, default=2048)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--sync_bn", action="store_true", help="BN statistics of the global batch (all-reduced)")
    p.add_argument("--no_augment", action="store_true", help="skip rotate + jitter (tests / debugging)")
    p.add_argument("--deterministic", action="store_true",
                   help="bit-reproducible backward passes: ordered owner sums instead of float atomics (slower)")
    return p.parse_args(argv)


def _load(path, with_mask, num, n_pts, seed, args):
    """-> (data (K,n,3) f32 | list of ragged clouds, labels, mask | None) on the host, NOT yet centred / normalised"""
    if with_mask == "parts":            # part segmentation: per-point part ids 0..5 (`train_partseg.py:93-94`)
        if path:
            return data_utils.load_npz(path, "parts") if path.endswith(".npz") else data_utils.load_parts_h5(path)
        data = synth_clouds(num, max(n_pts, 2048), seed=seed)
        parts = (np.floor((data[:, :, 1] + 1.0) * 3.0).clip(0, 5)).astype(np.int32)   # six height bands
        return data, synth_labels(num, seed), parts
    if path:
        if path.endswith(".npz"):
            arrs = data_utils.load_npz(path, with_mask)
        elif ".h5" in path:                                     # train.py:91-99
            arrs = data_utils.load_withmask_h5(path) if with_mask else data_utils.load_h5(path)
        else:                                                   # pickled split list + raw .bin objects
            if with_mask:
                raise ValueError("%s: pickled split lists of raw .bin objects carry no per-point masks; the mask / part "
                                 "models need an .h5 / .npz file with a 'mask' / 'parts' array" % path)
            arrs = data_utils.load_data(path, n_pts, with_bg_pl=args.with_bg, data_path=args.data_path)
        if with_mask:
            return arrs[0], arrs[1], data_utils.convert_to_binary_mask(arrs[2])
        return arrs[0], arrs[1], None
    data = synth_clouds(num, max(n_pts, 2048), seed=seed)
    return data, synth_labels(num, seed), synth_masks(num, data.shape[1], seed) if with_mask else None


def prepare_set(data, labels, mask, args, dev):
    """upload once; centre + normalise ON THE DEVICE exactly as `train.py:100-106` does on the host.  Ragged sets
    (raw .bin objects) are preprocessed per cloud on the host, as the reference's in-place loops do."""
    if isinstance(data, list):
        if args.center_data:
            data = [pc - pc.mean(axis=0, dtype=np.float32) for pc in data]
        if args.norm:
            data = [pc / np.sqrt((pc * pc).sum(axis=-1, dtype=np.float32)).max() for pc in data]
        return data, np.asarray(labels), mask
    x = torch.as_tensor(np.ascontiguousarray(data, dtype=np.float32), device=dev)
    if args.center_data:
        x = data_utils.center_data_device(x)
    if args.norm:
        x = data_utils.normalize_data_device(x)
    lab = torch.as_tensor(np.asarray(labels).reshape(-1).astype(np.int64), device=dev)
    m = None
    if mask is not None:
        mask = np.asarray(mask)
        m = torch.as_tensor((mask.reshape(mask.shape[0], -1) if mask.ndim == 3 else mask).astype(np.int64), device=dev)
    return x, lab, m


def epoch_view(data, labels, mask, num_point, rng, dev):
    """this epoch's (clouds, labels, mask) on the device: ONE random point subset shared by all clouds + a random
    cloud order (`data_utils.get_current_data_h5 / _withmask_h5 / _parts_h5`), as an index gather on the resident set"""
    if isinstance(data, list):                                  # ragged: per-cloud subsets on the host, then upload
        cur, lab = data_utils.get_current_data(data, labels, num_point, rng=rng)
        return (torch.as_tensor(cur, dtype=torch.float32, device=dev),
                torch.as_tensor(lab.astype(np.int64), device=dev), None)
    idx_pts, idx = data_utils.epoch_indices(data.shape[0], data.shape[1], num_point, rng)
    ip = torch.as_tensor(idx_pts, device=dev)
    ic = torch.as_tensor(idx, device=dev)
    cur = data.index_select(1, ip).index_select(0, ic)
    msk = mask.index_select(1, ip).index_select(0, ic) if mask is not None else None
    return cur, labels.index_select(0, ic), msk


def train(args):
    rank, world, local = D.init_from_env()
    if world == 1:
        if not 0 <= args.gpu < torch.cuda.device_count():
            raise RuntimeError("--gpu %d: no such device (%d visible)" % (args.gpu, torch.cuda.device_count()))
        local = args.gpu
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    D.SYNC_BN = bool(args.sync_bn)
    if args.deterministic:
        from scanobjectnn_amd import _lib
        _lib.set_deterministic(True)
    mod = importlib.import_module(MODELS[args.model])
    partseg = args.model.endswith("_partseg")
    with_mask = "parts" if partseg else args.model.endswith("_bga")
    per_rank = args.batch_size // world
    rng = np.random.RandomState(args.seed)          # same stream on every rank -> same epoch order
    gen = torch.Generator(device=dev)
    gen.manual_seed(args.seed + 1000 + rank)
    train_data, train_lab, train_mask = prepare_set(
        *_load(args.train_file, with_mask, args.synthetic_clouds, args.num_point, 1, args), args, dev)
    test_data, test_lab, test_mask = prepare_set(
        *_load(args.test_file, with_mask, max(args.batch_size, 256), args.num_point, 2, args), args, dev)

    example = torch.zeros((2, args.num_point, 3), device=dev)
    kw = {"num_class": args.num_class} if args.num_class != 15 else {}
    net = Model(mod.get_model, device=dev, seed=args.seed, **kw).build(example)
    fp = TU.FlatParams(net).enable_overlap(world)        # N > 1: gradient ranges travel while the backward pass runs
    D.broadcast_(fp.flat)
    opt = TU.make_optimizer(args.optimizer, fp, args.momentum)
    step = 0
    if rank == 0:
        os.makedirs(args.log_dir, exist_ok=True)
    log = []
    lo, hi = D.shard_range(args.batch_size, rank, world)
    for epoch in range(args.max_epoch):
        cur, lab, msk = epoch_view(train_data, train_lab, train_mask, args.num_point, rng, dev)
        nb = cur.shape[0] // args.batch_size
        t0 = time.time()
        tot = torch.zeros(3, dtype=torch.float64, device=dev)      # loss sum, correct, seen -- read once per epoch
        for b in range(nb):
            sl = slice(b * args.batch_size + lo, b * args.batch_size + hi)
            x = cur[sl]
            if not args.no_augment:
                x = provider.jitter_point_cloud(provider.rotate_point_cloud(x, generator=gen), generator=gen)
            x = x.contiguous()
            y = lab[sl]
            lr = TU.get_learning_rate(step, args.batch_size, args.learning_rate, args.decay_step, args.decay_rate)
            bn_decay = TU.get_bn_decay(step, args.batch_size, float(args.decay_step))
            fp.begin_step()
            out = net(x, is_training=True, bn_decay=bn_decay)
            if partseg:
                m = msk[sl]
                loss = mod.get_loss(out, m)
            elif with_mask:
                m = msk[sl]
                loss = mod.get_loss(out[0], out[1], y, m, seg_weight=args.seg_weight)[0]
            else:
                loss = mod.get_loss(out[0], y, out[1])
            loss.backward()
            fp.collect_mean(world)
            opt.step(lr)
            step += 1
            with torch.no_grad():
                tot[0] += loss.detach()
                if partseg:                      # point accuracy (`train_partseg.py:237-241`)
                    tot[1] += (out.argmax(dim=2) == m).sum()
                    tot[2] += per_rank * args.num_point
                else:
                    tot[1] += (out[0].argmax(dim=1) == y).sum()
                    tot[2] += per_rank
        if D.dist.is_initialized() and world > 1:
            D.dist.all_reduce(tot)
            tot[0] /= world
        loss_sum, correct, seen = (float(v) for v in tot.tolist())            # the epoch's only host sync
        D.broadcast_buffers_(net)                # per-rank BN moving statistics -> rank 0's, before eval / checkpoint
        # the reference draws a FRESH point subset and cloud order for every evaluation from the same global stream
        # as the training epochs (`train.py:275`: get_current_data_h5 on TEST_DATA) -- same here, same `rng`
        td, tl, tm = epoch_view(test_data, test_lab, test_mask, args.num_point, rng, dev)
        if partseg:
            ev = EV.eval_partseg_one_epoch(net, td, tm, per_rank, device=dev)
        elif with_mask:
            ev = EV.eval_seg_one_epoch(net, td, tl, tm, per_rank, device=dev)
        else:
            ev = EV.eval_one_epoch(net, td, tl, per_rank, device=dev, num_classes=args.num_class)
        rec = {"epoch": epoch, "mean_loss": loss_sum / max(nb, 1), "train_acc": correct / max(seen, 1),
               "eval_acc": ev["accuracy"], "eval_avg_class_acc": ev["avg_class_acc"],
               "clouds_per_s": nb * args.batch_size / (time.time() - t0)}
        if "seg_accuracy" in ev:                 # BGA models: the mask accuracy of `train_seg.py:330`
            rec["eval_seg_acc"] = ev["seg_accuracy"]
        log.append(rec)
        if rank == 0:
            print(json.dumps(rec), flush=True)
            torch.save(net.state_dict(), os.path.join(args.log_dir, "model.pt"))
    return log


if __name__ == "__main__":
    train(parse_args())
