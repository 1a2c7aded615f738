"""Evaluation loop -- restates the metric loop of `pointnet2/evaluate_scenennobjects.py:152-231`:
`num_votes` rotations about the up axis (angle vote/num_votes * 2π, provider.rotate_point_cloud_by_angle),
logits SUMMED over the votes, argmax, overall accuracy and mean per-class accuracy.  Visual dumps are out of
scope.  `eval_seg_one_epoch` adds the mask accuracy of `evaluate_seg_scenennobjects.py:336`
(correct points / (seen clouds * points))."""
import math

import numpy as np
import torch

from .. import provider

NUM_CLASSES = 15


def _host(a):
    """labels / masks may live on the device (the trainer keeps the whole set resident): metrics are host NumPy"""
    return a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)


def vote_logits(predict, points, num_votes):
    """sum over votes of predict(rotated points); predict: (B,N,3) tensor -> (B,C) logits"""
    total = None
    for vote_idx in range(num_votes):
        rotated = provider.rotate_point_cloud_by_angle(points, vote_idx / float(num_votes) * math.pi * 2)
        out = predict(rotated)
        total = out if total is None else total + out
    return total


def accuracy_summary(pred, labels, num_classes=NUM_CLASSES):
    """overall accuracy, mean per-class accuracy (classes never seen count as NaN -> ignored like np.mean would not;
    the reference divides by zero there), per-class vector"""
    pred, labels = np.asarray(pred), np.asarray(labels)
    seen = np.bincount(labels, minlength=num_classes).astype(np.float64)
    correct = np.bincount(labels[pred == labels], minlength=num_classes).astype(np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        per_class = correct / seen
    return float((pred == labels).mean()), float(np.nanmean(per_class)), per_class


@torch.no_grad()
def eval_one_epoch(net, data, labels, batch_size, num_votes=1, device="cuda:0"):
    """net: graph.Model of a classifier get_model; data (K,N,3), labels (K,).  Whole batches only (like the
    reference: num_batches = K // BATCH_SIZE)."""
    preds, seen = [], []
    labels = _host(labels)
    for b in range(data.shape[0] // batch_size):
        pts = torch.as_tensor(data[b * batch_size:(b + 1) * batch_size], dtype=torch.float32, device=device)
        logits = vote_logits(lambda p: net(p.contiguous(), is_training=False)[0], pts, num_votes)
        preds.append(logits.argmax(dim=1).cpu().numpy())
        seen.append(np.asarray(labels[b * batch_size:(b + 1) * batch_size]))
    pred, lab = np.concatenate(preds), np.concatenate(seen)
    acc, mean_class_acc, per_class = accuracy_summary(pred, lab)
    return {"accuracy": acc, "avg_class_acc": mean_class_acc, "per_class": per_class, "pred": pred, "label": lab}


@torch.no_grad()
def eval_seg_one_epoch(net, data, labels, masks, batch_size, device="cuda:0"):
    """BGA models: class accuracy + mask accuracy = correct points / (seen clouds * points)"""
    cls_pred, seen, seg_correct, n_pts = [], [], 0, 0
    labels, masks = _host(labels), _host(masks)
    for b in range(data.shape[0] // batch_size):
        sl = slice(b * batch_size, (b + 1) * batch_size)
        pts = torch.as_tensor(data[sl], dtype=torch.float32, device=device)
        class_pred, seg_pred = net(pts.contiguous(), is_training=False)
        cls_pred.append(class_pred.argmax(dim=1).cpu().numpy())
        seen.append(np.asarray(labels[sl]))
        seg_correct += int((seg_pred.argmax(dim=2).cpu().numpy() == np.asarray(masks[sl])).sum())
        n_pts += masks[sl].size
    pred, lab = np.concatenate(cls_pred), np.concatenate(seen)
    acc, mean_class_acc, per_class = accuracy_summary(pred, lab)
    return {"accuracy": acc, "avg_class_acc": mean_class_acc, "per_class": per_class,
            "seg_accuracy": seg_correct / float(n_pts)}


@torch.no_grad()
def eval_partseg_one_epoch(net, data, parts, batch_size, num_classes=6, device="cuda:0"):
    """part segmentation (`train_partseg.py:251-303`, `evaluate_partseg.py`): point accuracy = correct points /
    seen points; avg class acc = mean over the part classes that occur of (correct points of the class / points of it)"""
    seen_c = np.zeros(num_classes, dtype=np.int64)
    corr_c = np.zeros(num_classes, dtype=np.int64)
    parts = _host(parts)
    for b in range(data.shape[0] // batch_size):
        sl = slice(b * batch_size, (b + 1) * batch_size)
        pts = torch.as_tensor(data[sl], dtype=torch.float32, device=device)
        pred = net(pts.contiguous(), is_training=False).argmax(dim=2).cpu().numpy()
        gt = np.asarray(parts[sl])
        for c in range(num_classes):
            seen_c[c] += int((gt == c).sum())
            corr_c[c] += int(((gt == c) & (pred == c)).sum())
    occurs = seen_c > 0
    per_class = np.where(occurs, corr_c / np.maximum(seen_c, 1), 0.0)
    return {"accuracy": float(corr_c.sum()) / max(int(seen_c.sum()), 1),
            "avg_class_acc": float(per_class[occurs].mean()) if occurs.any() else 0.0,
            "per_class": per_class.tolist()}
