"""ORACLE (test infrastructure, never shipped or measured as the product).

CPU restatement of the EfficientDet post-processing chain:
  anchors            effdet/anchors.py:175-188 (get_feat_sizes), :249-299 (Anchors boxes)
  top-k selection    effdet/bench.py:12-56 (_post_process)
  box decode / clip  effdet/anchors.py:51-92
  detections         effdet/anchors.py:95-172 (generate_detections; fork: score>0.01 filter,
                     IoU 0.3 for hard and soft NMS, no padding)
  soft-NMS           effdet/soft_nms.py:12-169
  hard NMS           torchvision.ops.boxes.batched_nms - third-party, absent from /root/reference,
                     version unpinned (call site effdet/anchors.py:33,150).  Its published algorithm
                     (coordinate-offset trick, stable descending sort, suppress IoU > thr) is
                     restated here: PARITY UNPINNED for hard NMS.

Pinned by tests/golden/{anchors,post_process,decode,soft_nms,generate_detections}_*.npz, which
tools/make_golden.py produced by importing the reference modules in the build container.

Tie rule for top-k: the reference calls torch.topk, whose order among equal values is
unspecified; this oracle (and the HIP kernel) break ties by the lower flat index first.
"""
import numpy as np
import torch


# ------------------------------------------------------------------ anchors
def feat_sizes(image_size, max_level):
    fs = [tuple(image_size)]
    for _ in range(max_level):
        h, w = fs[-1]
        fs.append(((h - 1) // 2 + 1, (w - 1) // 2 + 1))
    return fs


def anchor_boxes(min_level, max_level, num_scales, aspect_ratios, anchor_scale, image_size):
    """[N,4] float32 yxyx; level-major, then y, x, then (octave-major, aspect-minor)."""
    fs = feat_sizes(image_size, max_level)
    scales = anchor_scale if isinstance(anchor_scale, (list, tuple)) else [anchor_scale] * (max_level - min_level + 1)
    out = []
    for level in range(min_level, max_level + 1):
        sy = fs[0][0] // fs[level][0]
        sx = fs[0][1] // fs[level][1]
        per_cfg = []
        for octave in range(num_scales):
            for aspect in aspect_ratios:
                a_scale = scales[level - min_level]
                base_x = a_scale * sx * 2 ** (octave / float(num_scales))
                base_y = a_scale * sy * 2 ** (octave / float(num_scales))
                if isinstance(aspect, (list, tuple)):
                    ax, ay = aspect
                else:
                    ax = np.sqrt(aspect)
                    ay = 1.0 / ax
                hx = base_x * ax / 2.0
                hy = base_y * ay / 2.0
                xs = np.arange(sx / 2, image_size[1], sx)
                ys = np.arange(sy / 2, image_size[0], sy)
                xv, yv = np.meshgrid(xs, ys)
                xv = xv.reshape(-1)
                yv = yv.reshape(-1)
                per_cfg.append(np.stack([yv - hy, xv - hx, yv + hy, xv + hx], axis=1)[:, None, :])
        out.append(np.concatenate(per_cfg, axis=1).reshape(-1, 4))
    return torch.from_numpy(np.vstack(out)).float()


# ------------------------------------------------------------------ top-k
def post_process(cls_outputs, box_outputs, num_levels, num_classes, max_detection_points=5000):
    B = cls_outputs[0].shape[0]
    cls_all = torch.cat([cls_outputs[l].permute(0, 2, 3, 1).reshape(B, -1, num_classes) for l in range(num_levels)], 1)
    box_all = torch.cat([box_outputs[l].permute(0, 2, 3, 1).reshape(B, -1, 4) for l in range(num_levels)], 1)
    flat = cls_all.reshape(B, -1)
    # descending by value, ties by lower index: stable sort of the negated values
    order = torch.sort(flat, dim=1, descending=True, stable=True)[1][:, :max_detection_points]
    indices = order // num_classes
    classes = order % num_classes
    box_topk = torch.gather(box_all, 1, indices.unsqueeze(2).expand(-1, -1, 4))
    cls_topk = torch.gather(flat, 1, order).unsqueeze(2)
    return cls_topk, box_topk, indices, classes


# ------------------------------------------------------------------ decode
def decode_box_outputs(rel_codes, anchors, output_xyxy=False):
    ya = (anchors[:, 0] + anchors[:, 2]) / 2
    xa = (anchors[:, 1] + anchors[:, 3]) / 2
    ha = anchors[:, 2] - anchors[:, 0]
    wa = anchors[:, 3] - anchors[:, 1]
    ty, tx, th, tw = rel_codes.unbind(dim=1)
    w = torch.exp(tw) * wa
    h = torch.exp(th) * ha
    yc = ty * ha + ya
    xc = tx * wa + xa
    ymin, xmin, ymax, xmax = yc - h / 2., xc - w / 2., yc + h / 2., xc + w / 2.
    if output_xyxy:
        return torch.stack([xmin, ymin, xmax, ymax], dim=1)
    return torch.stack([ymin, xmin, ymax, xmax], dim=1)


def clip_boxes_xyxy(boxes, size):
    boxes = boxes.clamp(min=0)
    return boxes.min(torch.cat([size, size], dim=0))


# ------------------------------------------------------------------ NMS
def _iou_one_to_many(box, boxes):
    area1 = (box[2] - box[0]) * (box[3] - box[1])
    area2 = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    wh = (torch.min(box[2:], boxes[:, 2:]) - torch.max(box[:2], boxes[:, :2])).clamp(min=0)
    inter = wh[:, 0] * wh[:, 1]
    return torch.where(inter > 0, inter / (area1 + area2 - inter), torch.zeros(1, dtype=inter.dtype))


def soft_nms(boxes, scores, method_gaussian=True, sigma=0.5, iou_threshold=0.5, score_threshold=0.005,
             max_picks=None):
    """soft_nms.py:42-112.  `max_picks` stops early (the caller only keeps the first
    max_det_per_image picks, anchors.py:153); None runs to exhaustion like the reference."""
    boxes_r = boxes.clone()
    scores_r = scores.clone()
    idxs = torch.arange(scores.numel())
    out_i, out_s = [], []
    while scores_r.numel() > 0 and (max_picks is None or len(out_i) < max_picks):
        top = int(torch.argmax(scores_r))
        out_i.append(int(idxs[top]))
        out_s.append(scores_r[top].clone())
        iou = _iou_one_to_many(boxes_r[top], boxes_r)
        if method_gaussian:
            decay = torch.exp(-torch.pow(iou, 2) / sigma)
        else:
            decay = torch.where(iou > iou_threshold, 1 - iou, torch.ones_like(iou))
        scores_r = scores_r * decay
        keep = scores_r > score_threshold
        keep[top] = False
        boxes_r, scores_r, idxs = boxes_r[keep], scores_r[keep], idxs[keep]
    if not out_i:
        return torch.empty(0, dtype=torch.int64), torch.empty(0, dtype=torch.float32)
    return torch.tensor(out_i, dtype=torch.int64), torch.stack(out_s)


def _class_offset_boxes(boxes, idxs):
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + 1)
    return boxes + offsets[:, None]


def batched_soft_nms(boxes, scores, idxs, method_gaussian=True, sigma=0.5, iou_threshold=0.5,
                     score_threshold=0.001, max_picks=None):
    if boxes.numel() == 0:
        return torch.empty(0, dtype=torch.int64), torch.empty(0, dtype=torch.float32)
    return soft_nms(_class_offset_boxes(boxes, idxs), scores, method_gaussian, sigma, iou_threshold,
                    score_threshold, max_picks)


def nms(boxes, scores, iou_threshold):
    """torchvision CPU nms: stable descending sort, greedy, suppress when IoU > threshold
    (float IoU compared against the double threshold)."""
    n = boxes.shape[0]
    if n == 0:
        return torch.empty(0, dtype=torch.int64)
    order = torch.sort(scores, descending=True, stable=True)[1]
    b = boxes[order]
    x1, y1, x2, y2 = b.unbind(1)
    areas = (x2 - x1) * (y2 - y1)
    suppressed = torch.zeros(n, dtype=torch.bool)
    keep = []
    thr = torch.tensor(iou_threshold, dtype=torch.float64)
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        xx1 = torch.maximum(x1[i], x1[i + 1:])
        yy1 = torch.maximum(y1[i], y1[i + 1:])
        xx2 = torch.minimum(x2[i], x2[i + 1:])
        yy2 = torch.minimum(y2[i], y2[i + 1:])
        w = (xx2 - xx1).clamp(min=0)
        h = (yy2 - yy1).clamp(min=0)
        inter = w * h
        ovr = inter / (areas[i] + areas[i + 1:] - inter)
        suppressed[i + 1:] |= ovr.double() > thr
    return order[torch.tensor(keep, dtype=torch.int64)]


def batched_nms(boxes, scores, idxs, iou_threshold):
    if boxes.numel() == 0:
        return torch.empty(0, dtype=torch.int64)
    return nms(_class_offset_boxes(boxes, idxs), scores, iou_threshold)


# ------------------------------------------------------------------ detections
def generate_detections(cls_outputs, box_outputs, anchors, indices, classes, img_scale, img_size,
                        max_det_per_image=100, soft=False, return_aux=False):
    a = anchors[indices, :]
    boxes = decode_box_outputs(box_outputs.float(), a, output_xyxy=True)
    if img_scale is not None and img_size is not None:
        boxes = clip_boxes_xyxy(boxes, img_size / img_scale)
    scores = cls_outputs.sigmoid().squeeze(1).float()
    m = scores > 0.01
    src = torch.nonzero(m).squeeze(1)          # position in the top-k list of every surviving candidate
    boxes, scores, classes = boxes[m], scores[m].clone(), classes[m]
    if soft:
        top, soft_scores = batched_soft_nms(boxes, scores, classes, True, 0.5, 0.3, 0.001,
                                            max_picks=max_det_per_image)
        scores[top] = soft_scores
    else:
        top = batched_nms(boxes, scores, classes, 0.3)
    top = top[:max_det_per_image]
    boxes = boxes[top]
    scores = scores[top, None]
    cls = classes[top, None] + 1
    if img_scale is not None:
        boxes = boxes * img_scale
    det = torch.cat([boxes, scores, cls.float()], dim=1)
    if return_aux:
        return det, src[top]
    return det


def auroc(pos, neg):
    """AUROC with `pos` as the positive class: P(pos > neg) + P(pos == neg)/2 (Mann-Whitney U / (n_p n_n)).
    Build-defined metric; pinned in tests against sklearn.metrics.roc_auc_score."""
    import numpy as np
    p = np.asarray(pos, dtype=np.float64).reshape(-1, 1)
    n = np.asarray(neg, dtype=np.float64).reshape(1, -1)
    return float(((p > n).sum() + 0.5 * (p == n).sum()) / (p.size * n.size))


def image_ood_score(energy):
    """[B, N] per-anchor energies -> [B] max_a(-energy_a)."""
    import numpy as np
    return (-np.asarray(energy, dtype=np.float32)).max(axis=1)


def novelty_score(proj_embds, confs, proto_idx, dot_mult, dot_add, sim_target='avg'):
    """infer.py:425-427 / 607-616 verbatim in meaning: proj = F.normalize(embds, p=2); sim_mat = proj @ proj.t();
    soft_thresh = sigmoid(dot_mult * (confs + dot_add)); 'avg': soft_thresh * sim_mat[:, max_idxs].mean(1) (:469-471, :616);
    'max': soft_thresh * sim_mat[:, max_idxs].max(1) (:453, :612 without the loss-side target_clust factor)."""
    import torch.nn.functional as F
    proj = F.normalize(proj_embds.float(), p=2)
    sim_mat = torch.matmul(proj, proj.t())
    st = (dot_mult * (confs.float() + dot_add)).sigmoid()
    cols = sim_mat[:, proto_idx]
    sim = cols.mean(1) if sim_target == 'avg' else cols.max(1)[0]
    return st * sim, st, sim
