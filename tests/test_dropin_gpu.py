"""Drop-in check: the core of the reference's pretrain.py loop (lines 160-276), written the way the script writes it -
`effdet` imported by that name, stock `torch.optim.Adam`, `torch.nn.utils.clip_grad_norm_`, `loss_fn`, `_post_process`,
`generate_detections`, the evaluator's per-image API with numpy arrays - runs on this package unchanged."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_pretrain_loop_as_written():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'ood_object_detection_amd'))          # shadows the reference's effdet/
    try:
        for k in [k for k in sys.modules if k == 'effdet' or k.startswith('effdet.')]:
            del sys.modules[k]
        from effdet.anchors import Anchors, AnchorLabeler, generate_detections
        from effdet.bench import _post_process
        from effdet.config import get_efficientdet_config
        from effdet.efficientdet import EfficientDet
        from effdet.evaluation.detection_evaluator import ObjectDetectionEvaluator
        from effdet.loss import DetectionLoss

        torch.manual_seed(0)
        h = get_efficientdet_config('tf_efficientdet_d0')
        h.image_size = (128, 128)
        h.num_classes = 6
        h.alpha, h.box_loss_weight = 0.15, 50.0                                   # pretrain.py:60-62
        model = EfficientDet(h, pretrained_backbone=False)
        model_config = model.config
        loss_fn = DetectionLoss(model_config)                                     # pretrain.py:155
        imagenet_mean = torch.tensor([x * 255 for x in (0.485, 0.456, 0.406)], device='cuda').view(1, 3, 1, 1)
        imagenet_std = torch.tensor([x * 255 for x in (0.229, 0.224, 0.225)], device='cuda').view(1, 3, 1, 1)
        model.to('cuda')
        anchors = Anchors.from_config(model_config).to('cuda')                    # pretrain.py:164
        labeler = AnchorLabeler(anchors, model_config.num_classes, match_threshold=0.5)

        def set_bn_eval(module):                                                  # pretrain.py:169-171
            if isinstance(module, torch.nn.modules.batchnorm._BatchNorm):
                module.eval()
        model.backbone.apply(set_bn_eval)                                         # freeze_bb_bn
        meta_optimizer = torch.optim.Adam([{'params': model.backbone.parameters()}, {'params': model.fpn.parameters()},
                                           {'params': model.class_net.parameters()}, {'params': model.box_net.parameters()}], lr=1e-3)
        cats = [{'id': i + 1, 'name': 'c%d' % i} for i in range(model_config.num_classes)]
        evaluator = ObjectDetectionEvaluator(cats, evaluate_corlocs=True)
        g = torch.Generator().manual_seed(1)
        qry_imgs_u8 = torch.randint(0, 256, (2, 3, 128, 128), generator=g, dtype=torch.uint8)
        qry_bbox = [torch.tensor([[10., 12., 70., 90.], [40., 30., 120., 100.]]), torch.tensor([[5., 5., 60., 50.]])]
        qry_cls = [torch.tensor([3, 5]), torch.tensor([1])]
        losses, norms = [], []
        for it in range(4):
            meta_optimizer.zero_grad()
            evaluator.clear()
            cls_anchor, bbox_anchor, num_positives = labeler.batch_label_anchors([b.cuda() for b in qry_bbox], [c.cuda() for c in qry_cls])
            qry_imgs = (qry_imgs_u8.to('cuda').float() - imagenet_mean) / imagenet_std                   # pretrain.py:226
            with torch.set_grad_enabled(True):
                feats = model(qry_imgs, mode='bb')                                                        # :229
            class_out, box_out = model(feats, mode='fpn_and_head')                                        # :232
            qry_loss, qry_class_loss, qry_box_loss = loss_fn(class_out, box_out, cls_anchor, bbox_anchor, num_positives)
            qry_loss.backward()                                                                           # :236
            with torch.no_grad():
                class_out_post, box_out_post, indices, classes = _post_process(
                    class_out, box_out, num_levels=model_config.num_levels, num_classes=model_config.num_classes,
                    max_detection_points=model_config.max_detection_points)
                for b_ix in range(2):
                    detections = generate_detections(class_out_post[b_ix], box_out_post[b_ix], anchors.boxes, indices[b_ix],
                                                     classes[b_ix], None, 128, max_det_per_image=100, soft_nms=False).cpu().numpy()
                    evaluator.add_single_ground_truth_image_info(b_ix, {'bbox': qry_bbox[b_ix].numpy(), 'cls': qry_cls[b_ix].numpy()})
                    bboxes_yxyx = np.concatenate([detections[:, 1:2], detections[:, 0:1], detections[:, 3:4], detections[:, 2:3]], axis=1)
                    evaluator.add_single_detected_image_info(b_ix, {'bbox': bboxes_yxyx, 'scores': detections[:, 4], 'cls': detections[:, 5]})
                map_metrics = evaluator.evaluate([c['name'] for c in cats], None)
            iter_meta_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 10.)                     # :272
            meta_optimizer.step()                                                                         # :276
            losses.append(float(qry_loss.detach()))
            norms.append(float(iter_meta_norm))
            assert 'Precision/mAP@0.5IOU' in map_metrics and 'Precision/meanCorLoc@0.5IOU' in map_metrics
        assert all(np.isfinite(losses)) and all(np.isfinite(norms)) and norms[0] > 0
        assert losses[-1] < losses[0], losses
        assert all(p.grad is not None for p in model.parameters())
    finally:
        sys.path.remove(os.path.join(root, 'ood_object_detection_amd'))
        for k in [k for k in sys.modules if k == 'effdet' or k.startswith('effdet.')]:
            del sys.modules[k]
