"""Training loop -- restates the step semantics of `pointnet2/train.py:136-171,218-261` (`train_seg.py` for the BGA
models, `train_partseg.py` for part segmentation) with the reference's flag names, data-parallel over the GPUs of
one node.

  python -m scanobjectnn_amd.pointnet2.train --model pointnet2_cls_ssg --num_point 2048 --batch_size 256 \
         --max_epoch 1 [--train_file x.npz --test_file y.npz]          (synthetic clouds when no file is given)
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m scanobjectnn_amd.pointnet2.train ...

Per step: device-side rotate + jitter (provider.py) -> forward (training BN, dropout) -> loss -> backward ->
ONE all-reduce of the flat gradient bucket -> TF-Adam with the staircase lr; bn_decay follows the reference
schedule; a checkpoint (`model.pt`, variables under the reference's TF scope names) is written every epoch.
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


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--model", default="pointnet2_cls_ssg", choices=sorted(MODELS))
    p.add_argument("--log_dir", default="log")
    p.add_argument("--num_point", type=int, default=1024)           # train.py:28
    p.add_argument("--max_epoch", type=int, default=250)            # :29
    p.add_argument("--batch_size", type=int, default=16, help="GLOBAL batch, split over the ranks")
    p.add_argument("--learning_rate", type=float, default=0.001)
    p.add_argument("--decay_step", type=int, default=200000)
    p.add_argument("--decay_rate", type=float, default=0.7)
    p.add_argument("--seg_weight", type=float, default=0.5)         # train_seg.py:35 (typed properly)
    p.add_argument("--train_file", default="")
    p.add_argument("--test_file", default="")
    p.add_argument("--synthetic_clouds", type=int, default=2048)
    p.add_argument("--seed", type=int, default=0)
    return p.parse_args(argv)


def _load(path, with_mask, num, n_pts, seed):
    if with_mask == "parts":            # part segmentation: per-point part ids 0..5 (`train_partseg.py:93-94`)
        if path:
            return data_utils.load_npz(path, "parts") if path.endswith(".npz") else data_utils.load_parts_h5(path)
        data = synth_clouds(num, max(n_pts, 2048), seed=seed)
        parts = (np.floor((data[:, :, 1] + 1.0) * 3.0).clip(0, 5)).astype(np.int32)   # six height bands
        return data, synth_labels(num, seed), parts
    if path:
        arrs = data_utils.load_npz(path, with_mask) if path.endswith(".npz") else \
            (data_utils.load_withmask_h5(path) if with_mask else data_utils.load_h5(path))
        if with_mask:
            return arrs[0], arrs[1], data_utils.convert_to_binary_mask(arrs[2])
        return arrs[0], arrs[1], None
    data = synth_clouds(num, max(n_pts, 2048), seed=seed)
    return data, synth_labels(num, seed), synth_masks(num, data.shape[1], seed) if with_mask else None


def train(args):
    rank, world, local = D.init_from_env()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    mod = importlib.import_module(MODELS[args.model])
    partseg = args.model.endswith("_partseg")
    with_mask = "parts" if partseg else args.model.endswith("_bga")
    per_rank = args.batch_size // world
    rng = np.random.RandomState(args.seed)          # same stream on every rank -> same epoch order
    gen = torch.Generator(device=dev)
    gen.manual_seed(args.seed + 1000 + rank)
    train_data, train_lab, train_mask = _load(args.train_file, with_mask, args.synthetic_clouds, args.num_point, 1)
    test_data, test_lab, test_mask = _load(args.test_file, with_mask, max(args.batch_size, 256), args.num_point, 2)

    example = torch.zeros((2, args.num_point, 3), device=dev)
    net = Model(mod.get_model, device=dev, seed=args.seed).build(example)
    fp = TU.FlatParams(net)
    D.broadcast_(fp.flat)
    opt = TU.TFAdam(fp)
    step = 0
    if rank == 0:
        os.makedirs(args.log_dir, exist_ok=True)
    log = []
    for epoch in range(args.max_epoch):
        if partseg:
            cur, lab, msk = data_utils.get_current_data_parts_h5(train_data, train_lab, np.squeeze(train_mask),
                                                                 args.num_point, rng=rng)
        elif with_mask:
            cur, lab, msk = data_utils.get_current_data_withmask_h5(train_data, train_lab, train_mask, args.num_point, rng=rng)
        else:
            (cur, lab), msk = data_utils.get_current_data_h5(train_data, train_lab, args.num_point, rng=rng), None
        nb = cur.shape[0] // args.batch_size
        t0, loss_sum, correct, seen = time.time(), 0.0, 0, 0
        for b in range(nb):
            lo, hi = D.shard_range(args.batch_size, rank, world)
            sl = slice(b * args.batch_size + lo, b * args.batch_size + hi)
            x = torch.as_tensor(cur[sl], dtype=torch.float32, device=dev)
            x = provider.jitter_point_cloud(provider.rotate_point_cloud(x, generator=gen), generator=gen).contiguous()
            y = torch.as_tensor(lab[sl], device=dev)
            lr = TU.get_learning_rate(step, args.batch_size, args.learning_rate, args.decay_step, args.decay_rate)
            bn_decay = TU.get_bn_decay(step, args.batch_size, float(args.decay_step))
            fp.begin_step()
            out = net(x, is_training=True, bn_decay=bn_decay)
            if partseg:
                m = torch.as_tensor(msk[sl], device=dev)
                loss = mod.get_loss(out, m)
            elif with_mask:
                m = torch.as_tensor(msk[sl], device=dev)
                loss = mod.get_loss(out[0], out[1], y, m, seg_weight=args.seg_weight)[0]
            else:
                loss = mod.get_loss(out[0], y, out[1])
            loss.backward()
            D.allreduce_mean_(fp.collect(), world)
            opt.step(lr)
            step += 1
            loss_sum += float(loss)
            if partseg:                      # point accuracy (`train_partseg.py:237-241`)
                correct += int((out.argmax(dim=2) == m).sum())
                seen += per_rank * args.num_point
            else:
                correct += int((out[0].argmax(dim=1) == y).sum())
                seen += per_rank
        if partseg:
            ev = EV.eval_partseg_one_epoch(net, test_data[:, :args.num_point], np.squeeze(test_mask)[:, :args.num_point],
                                           per_rank, device=dev)
        elif with_mask:
            ev = EV.eval_seg_one_epoch(net, test_data[:, :args.num_point], test_lab, test_mask[:, :args.num_point],
                                       per_rank, device=dev)
        else:
            ev = EV.eval_one_epoch(net, test_data[:, :args.num_point], test_lab, per_rank, device=dev)
        rec = {"epoch": epoch, "mean_loss": loss_sum / max(nb, 1), "train_acc": correct / max(seen, 1),
               "eval_acc": ev["accuracy"], "eval_avg_class_acc": ev["avg_class_acc"],
               "clouds_per_s": nb * args.batch_size / (time.time() - t0)}
        log.append(rec)
        if rank == 0:
            print(json.dumps(rec), flush=True)
            torch.save(net.state_dict(), os.path.join(args.log_dir, "model.pt"))
    return log


if __name__ == "__main__":
    train(parse_args())
