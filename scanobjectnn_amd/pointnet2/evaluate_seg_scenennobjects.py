"""Evaluation of the background-aware (BGA) models -- the caller `pointnet2/evaluate_seg_scenennobjects.py` restated
(the DGCNN variant `dgcnn/evaluate_seg_scenennobjects.py` is the same script on `dgcnn_bga`): its command line
(`:33-53`), the restore (`:172`), the vote loop (`:179-250`: `num_votes` rotations about the up axis at angle
vote/num_votes * 2 pi, class logits AND per-point mask logits summed over the votes, argmax of each), its metrics
(`:332-340`: mean loss, overall and mean per-class accuracy, `seg accuracy = correct points / (seen clouds * points)`,
the per-class table) and its outputs `dump_dir/pred_label.txt` (`:252`) and `dump_dir/log_evaluate.txt`.  The test set
is taken in file order with the FIRST `num_point` points of every cloud (`get_current_data_withmask_h5(...,
shuffle=False)`, `:196`); masks are binarised (`convert_to_binary_mask`, `:84`: background -1 -> 0, object parts -> 1).
Visual dumps (`--visu`, `--visu_mask`: obj / bin / jpg files of the masks, `:254-320`) are out of scope; the flags are
accepted.

  python -m scanobjectnn_amd.pointnet2.evaluate_seg_scenennobjects --model pointnet2_cls_bga --num_point 1024 \
         --batch_size 32 --model_path log/model.ckpt --test_file test_withmask.npz --num_votes 12

`--model_path`: this package's `model.pt`, or the prefix of a TensorFlow tensor bundle (as evaluate_scenennobjects).
"""
import argparse
import importlib
import math
import os

import numpy as np
import torch

from .. import data_utils, provider
from .evaluate_scenennobjects import SHAPE_NAMES, _host, accuracy_summary, restore

NUM_CLASSES = 15


def seg_summary(class_pred, labels, seg_pred, masks, num_classes=NUM_CLASSES):
    """the reference's end-of-epoch numbers (`:332-340`) from the per-cloud class predictions (K,), labels (K,), per-point
    mask predictions (K,N) and binary masks (K,N): overall / mean per-class accuracy and
    seg accuracy = correct points / (seen clouds * points per cloud)"""
    class_pred, labels = np.asarray(class_pred), np.asarray(labels)
    seg_pred, masks = np.asarray(seg_pred), np.asarray(masks)
    acc, mean_class_acc, per_class = accuracy_summary(class_pred, labels, num_classes)
    total_correct_seg = int((seg_pred == masks).sum())
    return {"accuracy": acc, "avg_class_acc": mean_class_acc, "per_class": per_class,
            "seg_accuracy": total_correct_seg / (float(len(labels)) * masks.shape[1])}


@torch.no_grad()
def eval_seg_votes(net, data, labels, masks, batch_size, num_votes=1, device="cuda:0", num_classes=NUM_CLASSES,
                   loss_fn=None):
    """net: graph.Model of a *_bga get_model -> (class_pred (B,C), seg_pred (B,N,2)); data (K,N,3), labels (K,), masks
    (K,N) in {0,1}.  Whole batches only (`num_batches = K // BATCH_SIZE`, `:203`).  loss_fn(class_pred, seg_pred, labels,
    masks) -> total loss of one vote; accumulated as loss * batch / num_votes (`:229`)."""
    labels, masks = _host(labels), _host(masks)
    cls_all, seg_all, seen_lab, seen_mask = [], [], [], []
    loss_sum = 0.0
    for b in range(data.shape[0] // batch_size):
        sl = slice(b * batch_size, (b + 1) * batch_size)
        pts = torch.as_tensor(data[sl], dtype=torch.float32, device=device)
        y = torch.as_tensor(np.asarray(labels[sl]).astype(np.int64), device=device)
        mk = torch.as_tensor(np.asarray(masks[sl]).astype(np.int64), device=device)
        cls_sum = seg_sum = None
        for vote_idx in range(num_votes):
            rotated = provider.rotate_point_cloud_by_angle(pts, vote_idx / float(num_votes) * math.pi * 2)
            class_pred, seg_pred = net(rotated.contiguous(), is_training=False)
            cls_sum = class_pred if cls_sum is None else cls_sum + class_pred
            seg_sum = seg_pred if seg_sum is None else seg_sum + seg_pred
            if loss_fn is not None:
                loss_sum += float(loss_fn(class_pred, seg_pred, y, mk)) * batch_size / float(num_votes)
        cls_all.append(cls_sum.argmax(dim=1).cpu().numpy())
        seg_all.append(seg_sum.argmax(dim=2).cpu().numpy())
        seen_lab.append(np.asarray(labels[sl]))
        seen_mask.append(np.asarray(masks[sl]))
    pred, lab = np.concatenate(cls_all), np.concatenate(seen_lab)
    seg, mask = np.concatenate(seg_all), np.concatenate(seen_mask)
    out = seg_summary(pred, lab, seg, mask, num_classes)
    out.update({"pred": pred, "label": lab, "seg_pred": seg, "mean_loss": loss_sum / max(len(lab), 1)})
    return out


def parse_args(argv=None):
    from .train import MODELS, _flag
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--gpu", type=int, default=0, help="GPU to use (:34)")
    p.add_argument("--model", default="pointnet2_cls_bga", choices=sorted(m for m in MODELS if m.endswith("_bga")))  # :35
    p.add_argument("--batch_size", type=int, default=1)                                       # :36
    p.add_argument("--num_point", type=int, default=1024)                                     # :37
    p.add_argument("--seg_weight", type=float, default=0.5, help="weight of the mask loss in the reported loss (:42)")
    p.add_argument("--model_path", default="log/model.ckpt",
                   help="model.pt of this package, or the prefix of a TensorFlow tensor bundle (:44)")
    p.add_argument("--dump_dir", default="dump/")                                             # :45
    p.add_argument("--with_bg", type=_flag, default=True)                                     # :46
    p.add_argument("--norm", type=_flag, default=True)                                        # :47
    p.add_argument("--center_data", type=_flag, default=True)                                 # :48
    p.add_argument("--test_file", default="", help=".npz / .h5 with data, label, mask (:50); synthetic clouds if empty")
    p.add_argument("--normal", action="store_true", help="accepted for CLI compatibility (:52); never read")
    p.add_argument("--num_votes", type=int, default=1)                                        # :53
    p.add_argument("--visu", type=_flag, default=False, help="accepted (:54); image dumps are out of scope")
    p.add_argument("--visu_mask", type=_flag, default=False, help="accepted (:55); mask dumps are out of scope")
    p.add_argument("--shape_names", default="", help="text file with one class name per line (default: the 15 "
                   "ScanObjectNN names; the reference reads ../training_data/shape_names_ext.txt, :76-77)")
    p.add_argument("--synthetic_clouds", type=int, default=256)
    return p.parse_args(argv)


def load_test_set(args):
    """(:83-92): load with masks, binarise, centre, normalise on the host exactly like the reference"""
    if not args.test_file:
        from ..synth import synth_clouds, synth_labels, synth_masks
        data = synth_clouds(args.synthetic_clouds, max(args.num_point, 2048), seed=2)
        labels = synth_labels(args.synthetic_clouds, 2)
        masks = synth_masks(args.synthetic_clouds, data.shape[1], 2)
    else:
        if args.test_file.endswith(".npz"):
            data, labels, masks = data_utils.load_npz(args.test_file, True)
        else:
            data, labels, masks = data_utils.load_withmask_h5(args.test_file)
        masks = data_utils.convert_to_binary_mask(masks)
    if args.center_data:
        data = data_utils.center_data(data)
    if args.norm:
        data = data_utils.normalize_data(data)
    return data, np.asarray(labels).reshape(-1), masks


def evaluate(args):
    from ..graph import Model
    from .train import MODELS
    if not torch.cuda.is_available() or not 0 <= args.gpu < torch.cuda.device_count():
        raise RuntimeError("--gpu %d: no such device (%d visible)" % (args.gpu, torch.cuda.device_count()))
    dev = torch.device("cuda", args.gpu)
    torch.cuda.set_device(dev)
    mod = importlib.import_module(MODELS[args.model])
    names = [l.rstrip() for l in open(args.shape_names)] if args.shape_names else SHAPE_NAMES
    os.makedirs(args.dump_dir, exist_ok=True)
    log_f = open(os.path.join(args.dump_dir, "log_evaluate.txt"), "w")
    log_f.write(str(args) + "\n")

    def log_string(s):
        log_f.write(s + "\n")
        log_f.flush()
        print(s)

    data, labels, masks = load_test_set(args)
    net = Model(mod.get_model, device=dev, seed=0).build(torch.zeros((2, args.num_point, 3), device=dev))
    info = restore(net, args.model_path)
    log_string("Model restored. (%s, %d variables)" % (info["format"], info["loaded"]))
    # file order, the first num_point points of every cloud (:196, shuffle=False)
    cur, lab, msk = data_utils.get_current_data_withmask_h5(data, labels, masks, args.num_point, shuffle=False)
    lab, msk = np.squeeze(lab), np.squeeze(msk)
    ev = eval_seg_votes(net, cur, lab, msk, args.batch_size, num_votes=args.num_votes, device=dev,
                        loss_fn=lambda cp, sp, y, mk: mod.get_loss(cp, sp, y, mk, seg_weight=args.seg_weight)[0])
    with open(os.path.join(args.dump_dir, "pred_label.txt"), "w") as fout:                 # :252
        for p, l in zip(ev["pred"], ev["label"]):
            fout.write("%s, %s\n" % (names[p], names[l]))
    log_string("total seen: %d" % len(ev["label"]))
    log_string("eval mean loss: %f" % ev["mean_loss"])
    log_string("eval accuracy: %f" % ev["accuracy"])
    log_string("eval avg class acc: %f" % ev["avg_class_acc"])
    log_string("seg accuracy: %f" % ev["seg_accuracy"])                                    # :336
    for i, name in enumerate(names[:NUM_CLASSES]):
        log_string("%10s:\t%0.3f" % (name, ev["per_class"][i]))
    log_f.close()
    return ev


if __name__ == "__main__":
    evaluate(parse_args())
