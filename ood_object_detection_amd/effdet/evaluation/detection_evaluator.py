"""PASCAL-style detection evaluation on the device (reference: effdet/evaluation/detection_evaluator.py:96-316 on top of
object_detection_evaluation.py, per_image_evaluation.py, metrics.py - ~3000 lines of per-image numpy in the fork, run
inside its training loop, pretrain.py:246-252).

Same surface as the reference class for the inputs that loop produces (boxes, scores, 1-based classes; no difficult /
group-of flags, no masks - those raise): `add_single_ground_truth_image_info`, `add_single_detected_image_info`,
`evaluate(task_categories, batch_cats=None)` with the reference's metric names, `clear()`.  `add_batch` is the native
entry: the `[B, max_det, 6]` detections + counts of `DetBenchPredict` / `batched_detections` and padded ground truth go
to ONE launch (`effdet_eval_match`, a workgroup per image); `evaluate` runs `effdet_eval_ap` over everything
accumulated.  State lives on the GPU; only the final per-class vectors are copied to the host.  No CPU fallback.
"""
import numpy as np
import torch

from ... import _lib


def create_category_index(categories):
    return {cat['id']: cat for cat in categories}


class ObjectDetectionEvaluator(object):
    def __init__(self, categories, matching_iou_threshold=0.5, recall_lower_bound=0.0, recall_upper_bound=1.0,
                 evaluate_corlocs=False, evaluate_precision_recall=False, metric_prefix=None, use_weighted_mean_ap=False,
                 evaluate_masks=False, group_of_weight=0.0, device='cuda:0'):
        if evaluate_masks or use_weighted_mean_ap or evaluate_precision_recall or group_of_weight != 0.0 \
                or recall_lower_bound != 0.0 or recall_upper_bound != 1.0:
            raise NotImplementedError('only the box-mode PASCAL metrics the fork uses are built (mAP, per-class AP, CorLoc)')
        self._categories = categories
        self._num_classes = max(cat['id'] for cat in categories)
        if min(cat['id'] for cat in categories) < 1:
            raise ValueError('Classes should be 1-indexed.')
        self._matching_iou_threshold = float(matching_iou_threshold)
        self._label_id_offset = 1
        self._evaluate_corlocs = evaluate_corlocs
        self._metric_prefix = (metric_prefix + '_') if metric_prefix else ''
        self._metric_names = [self._metric_prefix + 'Precision/mAP@{}IOU'.format(self._matching_iou_threshold)]
        if evaluate_corlocs:
            self._metric_names.append(self._metric_prefix + 'Precision/meanCorLoc@{}IOU'.format(self._matching_iou_threshold))
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('the evaluator runs on the GPU (no CPU fallback)')
        self.lib = _lib.load()
        self.clear()

    # ---- state ------------------------------------------------------------------------------------------
    def clear(self):
        C, dev = self._num_classes, self.device
        self._gt_count = torch.zeros(C, dtype=torch.int32, device=dev)
        self._gt_imgs = torch.zeros(C, dtype=torch.int32, device=dev)
        self._correct = torch.zeros(C, dtype=torch.int32, device=dev)
        self._scores, self._classes, self._tp = [], [], []
        self._pending_gt = {}
        self._seen_det = set()
        self._image_ids = set()

    def _st(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    # ---- native entry -----------------------------------------------------------------------------------
    def add_batch(self, det, count, gt_boxes, gt_cls):
        """det [B, max_det, 6] rows x1,y1,x2,y2,score,class (1-based), sorted by descending score, `count[b]` valid rows;
        gt_boxes [B, M, 4] yxyx, gt_cls [B, M] 1-based (<= 0: padding)."""
        dev = self.device
        det = det.to(device=dev, dtype=torch.float32).contiguous()
        count = count.to(device=dev, dtype=torch.int32).contiguous()
        gt_boxes = gt_boxes.to(device=dev, dtype=torch.float32).contiguous()
        gt_cls = gt_cls.to(device=dev, dtype=torch.int64).contiguous()
        B, max_det, _ = det.shape
        M = gt_boxes.shape[1]
        if M == 0:
            gt_boxes = torch.zeros(B, 1, 4, dtype=torch.float32, device=dev)
            gt_cls = torch.zeros(B, 1, dtype=torch.int64, device=dev)
            M = 1
        tp = torch.empty(B, max_det, dtype=torch.int32, device=dev)
        _lib.check(self.lib.effdet_eval_match(self._st(), det.data_ptr(), count.data_ptr(), gt_boxes.data_ptr(), gt_cls.data_ptr(),
                                              B, max_det, M, self._num_classes, self._matching_iou_threshold, tp.data_ptr(),
                                              self._gt_count.data_ptr(), self._gt_imgs.data_ptr(), self._correct.data_ptr()),
                   'effdet_eval_match')
        self._scores.append(det[:, :, 4].reshape(-1))
        self._classes.append((det[:, :, 5].to(torch.int32) - 1).reshape(-1))
        self._tp.append(tp.reshape(-1))
        return tp

    # ---- reference API ----------------------------------------------------------------------------------
    def add_single_ground_truth_image_info(self, image_id, gt_dict):
        if image_id in self._image_ids:
            return
        for k in gt_dict:
            if k in ('difficult', 'group_of', 'instance_masks', 'groundtruth_difficult', 'groundtruth_group_of'):
                raise NotImplementedError('difficult / group-of / mask ground truth is not built')
        self._pending_gt[image_id] = (torch.as_tensor(np.asarray(gt_dict['bbox']), dtype=torch.float32).reshape(-1, 4),
                                      torch.as_tensor(np.asarray(gt_dict['cls']), dtype=torch.int64).reshape(-1))
        self._image_ids.add(image_id)

    def add_single_detected_image_info(self, image_id, detections_dict):
        """detections_dict: 'bbox' [n,4] yxyx, 'scores' [n], 'cls' [n] 1-based (pretrain.py:248-250)"""
        if image_id in self._seen_det:
            return
        self._seen_det.add(image_id)
        box = torch.as_tensor(np.asarray(detections_dict['bbox']), dtype=torch.float32).reshape(-1, 4)
        sc = torch.as_tensor(np.asarray(detections_dict['scores']), dtype=torch.float32).reshape(-1)
        cl = torch.as_tensor(np.asarray(detections_dict['cls'])).reshape(-1).to(torch.float32)
        order = torch.argsort(sc, descending=True, stable=True)          # index plumbing; the matching needs score order
        n = box.shape[0]
        det = torch.zeros(1, max(n, 1), 6)
        if n:
            b = box[order]
            det[0, :n] = torch.stack([b[:, 1], b[:, 0], b[:, 3], b[:, 2], sc[order], cl[order]], 1)
        gt_b, gt_c = self._pending_gt.pop(image_id, (torch.zeros(0, 4), torch.zeros(0, dtype=torch.int64)))
        self.add_batch(det, torch.tensor([n], dtype=torch.int32), gt_b[None], gt_c[None])

    def _flush_pending(self):
        """ground truth of images that never received detections still counts (the reference counts it when it is added,
        object_detection_evaluation.py:87-139)"""
        for image_id in list(self._pending_gt.keys()):
            gt_b, gt_c = self._pending_gt.pop(image_id)
            self._seen_det.add(image_id)
            self.add_batch(torch.zeros(1, 1, 6), torch.zeros(1, dtype=torch.int32), gt_b[None], gt_c[None])

    def _compute(self):
        C, dev = self._num_classes, self.device
        self._flush_pending()
        ap = torch.full((C,), float('nan'), dtype=torch.float64, device=dev)
        n = sum(t.numel() for t in self._scores)
        if n > 65536:
            raise RuntimeError('%d detections accumulated; effdet_eval_ap handles 65536 between clear() calls (the training loop '
                               'clears every iteration, pretrain.py:223)' % n)
        if n:
            scores, classes, tp = torch.cat(self._scores).contiguous(), torch.cat(self._classes).contiguous(), torch.cat(self._tp).contiguous()
            nb = self.lib.effdet_eval_ap_workspace_bytes(n)
            ws = torch.empty((nb + 7) // 8, dtype=torch.float64, device=dev)
            _lib.check(self.lib.effdet_eval_ap(self._st(), scores.data_ptr(), classes.data_ptr(), tp.data_ptr(), n, C,
                                               self._gt_count.data_ptr(), ap.data_ptr(), ws.data_ptr(), ws.numel() * 8), 'effdet_eval_ap')
        ap = ap.cpu().numpy()
        gt_count = self._gt_count.cpu().numpy()
        if not n:
            ap = np.where(gt_count > 0, 0.0, np.nan)
        gt_imgs = self._gt_imgs.cpu().numpy()
        correct = self._correct.cpu().numpy().astype(float)
        with np.errstate(invalid='ignore', divide='ignore'):
            corloc = np.where(gt_imgs == 0, np.nan, correct / gt_imgs)
            mean_ap = float(np.nanmean(ap)) if np.isfinite(ap).any() else float('nan')
            mean_corloc = float(np.nanmean(corloc)) if np.isfinite(corloc).any() else float('nan')
        return dict(per_class_ap=ap, mean_ap=mean_ap, per_class_corlocs=corloc, mean_corloc=mean_corloc)

    def evaluate(self, task_categories=None, batch_cats=None):
        """Metric dict with the reference's keys (detection_evaluator.py:268-305)."""
        metrics = self._compute()
        out = {self._metric_names[0]: metrics['mean_ap']}
        if self._evaluate_corlocs:
            out[self._metric_names[1]] = metrics['mean_corloc']
        if task_categories is None:
            task_categories = [create_category_index(self._categories)[i + 1]['name'] for i in range(self._num_classes)]
        for idx, category_name in enumerate(task_categories):
            if batch_cats is not None and idx not in batch_cats:
                continue
            out['AP@{}IOU/{}'.format(self._matching_iou_threshold, category_name)] = metrics['per_class_ap'][idx]
            if self._evaluate_corlocs:
                out['CorLoc@{}IOU/{}'.format(self._matching_iou_threshold, category_name)] = metrics['per_class_corlocs'][idx]
        return out


class PascalDetectionEvaluator(ObjectDetectionEvaluator):
    """effdet/evaluation/detection_evaluator.py:317-326"""

    def __init__(self, categories, matching_iou_threshold=0.5, **kw):
        super().__init__(categories, matching_iou_threshold=matching_iou_threshold, evaluate_corlocs=False,
                         metric_prefix='PascalBoxes', **kw)
