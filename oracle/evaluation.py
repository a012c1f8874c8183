"""ORACLE (test infrastructure only - never imported by the product path).

numpy restatement of the PASCAL-style detection evaluation the fork runs inside its training loop
(pretrain.py:246-252: ObjectDetectionEvaluator(evaluate_corlocs=True), cleared every iteration), for the inputs that loop
produces: boxes + scores + classes only (no difficult / group-of flags, no masks).

Follows, by reference file:line
  effdet/evaluation/per_image_evaluation.py:514-538  _remove_invalid_boxes (ymin < ymax and xmin < xmax, strict)
  effdet/evaluation/per_image_evaluation.py:268-299  per class: detections sorted by descending score (the embedded
                                                     non_max_suppression runs with iou_threshold 1.0 = sort only)
  effdet/evaluation/per_image_evaluation.py:377-405  compute_match_iou: each detection takes the ground-truth box of its
                                                     class with the largest IoU (first one on ties); IoU >= threshold and box
                                                     not taken yet -> true positive and the box is taken; otherwise false positive
  effdet/evaluation/per_image_evaluation.py:143-176  CorLoc: the top-scoring detection of a class overlaps some ground-truth
                                                     box of that class with IoU >= threshold
  effdet/evaluation/np_box_ops.py                    area / intersection / iou in the dtype of the inputs (float32)
  effdet/evaluation/metrics.py:4-90                  compute_precision_recall, compute_average_precision (VOC all-points)
  effdet/evaluation/metrics.py:92-106                compute_cor_loc
  effdet/evaluation/object_detection_evaluation.py:205-273  per-class AP for classes with ground truth, nanmean -> mAP, mean CorLoc

Pinned: tests/golden/evaluation.npz was produced by the reference's own ObjectDetectionEvaluator (tools/make_golden.py).
"""
import numpy as np


def iou_matrix(a, b):
    """a [N,4], b [M,4] yxyx float32 -> [N,M] float32 (np_box_ops.iou)"""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    ih = np.maximum(np.float32(0), np.minimum(a[:, 2:3], b[:, 2][None]) - np.maximum(a[:, 0:1], b[:, 0][None]))
    iw = np.maximum(np.float32(0), np.minimum(a[:, 3:4], b[:, 3][None]) - np.maximum(a[:, 1:2], b[:, 1][None]))
    inter = ih * iw
    return inter / (area_a[:, None] + area_b[None] - inter)


def per_image(det_boxes, det_scores, det_classes, gt_boxes, gt_classes, num_classes, thr=0.5):
    """0-based classes, yxyx boxes.  -> (scores per class, tp labels per class, is_class_correctly_detected [C])"""
    det_boxes = np.asarray(det_boxes, np.float32).reshape(-1, 4)
    det_scores, det_classes = np.asarray(det_scores), np.asarray(det_classes)
    valid = np.logical_and(det_boxes[:, 0] < det_boxes[:, 2], det_boxes[:, 1] < det_boxes[:, 3])
    det_boxes, det_scores, det_classes = det_boxes[valid], det_scores[valid], det_classes[valid]
    gt_boxes = np.asarray(gt_boxes, np.float32).reshape(-1, 4)
    gt_classes = np.asarray(gt_classes)
    scores_c, tp_c = [], []
    correct = np.zeros(num_classes, dtype=int)
    for c in range(num_classes):
        d = det_boxes[det_classes == c]
        s = det_scores[det_classes == c]
        g = gt_boxes[gt_classes == c]
        if d.size > 0 and g.size > 0:
            top = np.argmax(s)
            if np.max(iou_matrix(d[top:top + 1], g)) >= thr:
                correct[c] = 1
        if d.size == 0:
            scores_c.append(np.array([], dtype=float))
            tp_c.append(np.array([], dtype=bool))
            continue
        order = np.argsort(s)[::-1]
        d, s = d[order], s[order]
        tp = np.zeros(len(s), dtype=bool)
        if g.size > 0:
            iou = iou_matrix(d, g)
            best = np.argmax(iou, axis=1)
            taken = np.zeros(len(g), dtype=bool)
            for i in range(len(s)):
                if iou[i, best[i]] >= thr and not taken[best[i]]:
                    tp[i] = True
                    taken[best[i]] = True
        scores_c.append(s)
        tp_c.append(tp)
    return scores_c, tp_c, correct


def average_precision(scores, tp, num_gt):
    """metrics.py:4-90 on one class; NaN when the class has no ground truth"""
    if num_gt == 0:
        return float('nan')
    scores, tp = np.asarray(scores, dtype=float), np.asarray(tp, dtype=float)
    if scores.size == 0:
        return 0.0
    order = np.argsort(scores)[::-1]
    t = tp[order]
    ctp = np.cumsum(t)
    cfp = np.cumsum((t <= 0).astype(float))
    precision = ctp / (ctp + cfp)
    recall = ctp / num_gt
    recall = np.concatenate([[0], recall, [1]])
    precision = np.concatenate([[0], precision, [0]])
    for i in range(len(precision) - 2, -1, -1):
        precision[i] = max(precision[i], precision[i + 1])
    idx = np.where(recall[1:] != recall[:-1])[0] + 1
    return float(np.sum((recall[idx] - recall[idx - 1]) * precision[idx]))


def evaluate(images, num_classes, thr=0.5):
    """images: list of dicts(det_boxes, det_scores, det_classes, gt_boxes, gt_classes), 0-based classes.
    -> dict(mean_ap, mean_corloc, per_class_ap [C], per_class_corloc [C])"""
    scores = [[] for _ in range(num_classes)]
    tps = [[] for _ in range(num_classes)]
    num_gt = np.zeros(num_classes)
    num_gt_imgs = np.zeros(num_classes, dtype=int)
    correct = np.zeros(num_classes)
    for im in images:
        gc = np.asarray(im['gt_classes'])
        for c in range(num_classes):
            n = int(np.sum(gc == c))
            num_gt[c] += n
            num_gt_imgs[c] += 1 if n > 0 else 0
        s, t, cor = per_image(im['det_boxes'], im['det_scores'], im['det_classes'], im['gt_boxes'], gc, num_classes, thr)
        for c in range(num_classes):
            if len(s[c]):
                scores[c].append(s[c])
                tps[c].append(t[c])
        correct += cor
    ap = np.full(num_classes, np.nan)
    for c in range(num_classes):
        if num_gt[c] == 0:
            continue
        sc = np.concatenate(scores[c]) if scores[c] else np.array([])
        tc = np.concatenate(tps[c]) if tps[c] else np.array([])
        ap[c] = average_precision(sc, tc, num_gt[c])
    with np.errstate(invalid='ignore', divide='ignore'):
        corloc = np.where(num_gt_imgs == 0, np.nan, correct / num_gt_imgs)
    return dict(mean_ap=float(np.nanmean(ap)), mean_corloc=float(np.nanmean(corloc)), per_class_ap=ap, per_class_corloc=corloc)
