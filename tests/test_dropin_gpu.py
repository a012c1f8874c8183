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


def test_infer_validation_iteration_as_written():
    """infer.py's model set-up (:173-206) and the forward-only part of an iteration (:341-351, :561-563, :681, :689-700):
    MetaHead swapped in from the class_net parameters, num_classes = 1, strict state-dict round trips, modes supp_bb / bb /
    not_cls / supp_cls / qry_cls with fast weights, _post_process + generate_detections + evaluator."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'ood_object_detection_amd'))
    try:
        for k in [k for k in sys.modules if k == 'effdet' or k.startswith('effdet.')]:
            del sys.modules[k]
        from effdet.anchors import Anchors, generate_detections
        from effdet.bench import _post_process
        from effdet.config import get_efficientdet_config
        from effdet.efficientdet import EfficientDet, MetaHead, ProjectionNet
        from effdet.evaluation.detection_evaluator import ObjectDetectionEvaluator
        from effdet.loss import DetectionLoss

        torch.manual_seed(0)
        h = get_efficientdet_config('tf_efficientdet_d0')
        h.image_size = (256, 256)          # 12 276 anchors x 1 class > max_detection_points (torch.topk needs k <= N*C, too)
        h.num_classes = 4
        model = EfficientDet(h, pretrained_backbone=False)
        state_dict = {k: v.clone() for k, v in model.state_dict().items()}
        model.load_state_dict(state_dict, strict=True)                                                # infer.py:185
        class_net_init_params = {n: v.data.detach().clone() for n, v in model.named_parameters() if 'class_net' in n}
        model.class_net = MetaHead(model.config, pretrain_init=class_net_init_params)                 # :191
        model.config.num_classes = 1                                                                  # :192
        model_config = model.config
        proj_net = ProjectionNet(model_config, 64)                                                    # :196
        model.load_state_dict({k: v.clone() for k, v in model.state_dict().items()}, strict=True)    # :199
        proj_net.load_state_dict(proj_net.state_dict(), strict=True)
        model.to('cuda').eval()
        anchors = Anchors.from_config(model_config).to('cuda')                                        # :206
        loss_fn = DetectionLoss(model_config)                                                         # :212
        assert loss_fn is not None
        g = torch.Generator().manual_seed(2)
        supp_imgs = torch.randn(3, 3, 256, 256, generator=g).to('cuda')
        qry_imgs = torch.randn(2, 3, 256, 256, generator=g).to('cuda')
        with torch.no_grad():
            supp_activs = model(supp_imgs, mode='supp_bb')                                            # :343
            qry_feats = model(qry_imgs, mode='bb')                                                    # :346
            qry_activs, qry_box_out = model(qry_feats, mode='not_cls')                                # :349
            qry_activs = [a.clone() for a in qry_activs]
            anch_confs, obj_embds = model(supp_activs, fast_weights=None, mode='supp_cls')            # :563
            # FLAGS.supp_level_offset defaults to 2 (infer.py:94): the support pass returns the 3 coarsest levels, lining up with
            # the 3-level proj_anchors of dataloader.py:66
            assert len(anch_confs) == model_config.num_levels - 2 and anch_confs[0].shape[:2] == (3, 9)
            assert obj_embds[0].shape[1] == model_config.fpn_channels
            fast_weights = [par - 0.01 * torch.ones_like(par) if 'predict_p' in n else par
                            for n, par in model.class_net.named_parameters()]                        # :660-678 (first-order stand-in)
            qry_class_out = model(qry_activs, fast_weights=fast_weights, mode='qry_cls')              # :681
            base_out = model(qry_activs, fast_weights=None, mode='qry_cls')
            assert not torch.equal(qry_class_out[0], base_out[0])                                     # the fast weights are used
            class_out_post, box_out_post, indices, classes = _post_process(
                qry_class_out, qry_box_out, num_levels=model_config.num_levels, num_classes=model_config.num_classes,
                max_detection_points=model_config.max_detection_points)                               # :690
            evaluator = ObjectDetectionEvaluator([{'id': 1, 'name': 'obj'}], evaluate_corlocs=True)
            gt = [np.array([[10., 12., 70., 90.]], np.float32), np.array([[5., 5., 60., 50.], [40., 30., 120., 100.]], np.float32)]
            for b_ix in range(2):
                detections = generate_detections(class_out_post[b_ix], box_out_post[b_ix], anchors.boxes, indices[b_ix], classes[b_ix],
                                                 None, 256, max_det_per_image=30, soft_nms=False).cpu().numpy()   # :694
                assert detections.shape[1] == 6 and detections.shape[0] <= 30
                evaluator.add_single_ground_truth_image_info(b_ix, {'bbox': gt[b_ix], 'cls': np.ones(len(gt[b_ix]), np.int64)})
                bboxes_yxyx = np.concatenate([detections[:, 1:2], detections[:, 0:1], detections[:, 3:4], detections[:, 2:3]], axis=1)
                evaluator.add_single_detected_image_info(b_ix, {'bbox': bboxes_yxyx, 'scores': detections[:, 4], 'cls': detections[:, 5]})
            map_metrics = evaluator.evaluate(['obj'])                                                 # :700
        assert 'Precision/mAP@0.5IOU' in map_metrics and 'AP@0.5IOU/obj' in map_metrics
    finally:
        sys.path.remove(os.path.join(root, 'ood_object_detection_amd'))
        for k in [k for k in sys.modules if k == 'effdet' or k.startswith('effdet.')]:
            del sys.modules[k]


def test_infer_training_iteration_inner_and_outer_gradients():
    """infer.py:561-685 as the script writes it: support pass through the MetaHead with grad, BCE on the anchor confidences,
    `torch.autograd.grad(..., model.class_net.parameters(), allow_unused=True, only_inputs=True, create_graph=True)` (:658),
    fast weights `par - par_lr * inner_grad` (:660-678), query pass `mode='qry_cls'` with them (:681), `loss_fn` (:683),
    `final_loss.backward()` (:687) - the outer gradient reaches the head's parameters and the learnable inner learning rates.
    (The inner gradient is itself differentiable - effdet/meta_ops.py - so this is the reference's second-order MAML; values are
    checked against the oracle in test_model_gpu.py::test_meta_head_second_order_matches_oracle_autograd.)"""
    import torch.nn.functional as F
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'ood_object_detection_amd'))
    try:
        for k in [k for k in sys.modules if k == 'effdet' or k.startswith('effdet.')]:
            del sys.modules[k]
        from effdet.config import get_efficientdet_config
        from effdet.efficientdet import EfficientDet, MetaHead
        from effdet.loss import DetectionLoss
        torch.manual_seed(0)
        h = get_efficientdet_config('tf_efficientdet_d0')
        h.image_size = (128, 128)
        h.num_classes = 4
        model = EfficientDet(h, pretrained_backbone=False)
        class_net_init_params = {n: v.data.detach().clone() for n, v in model.named_parameters() if 'class_net' in n}
        model.class_net = MetaHead(model.config, pretrain_init=class_net_init_params)                 # infer.py:191
        model.config.num_classes = 1                                                                  # :192
        model_config = model.config
        model.to('cuda')
        loss_fn = DetectionLoss(model_config)                                                         # :212
        learnable_lr = [torch.nn.Parameter(torch.tensor(0.1, device='cuda')) for _ in range(model_config.box_class_repeats + 2)]
        g = torch.Generator().manual_seed(3)
        supp_imgs = torch.randn(3, 3, 128, 128, generator=g).to('cuda')
        qry_imgs = torch.randn(2, 3, 128, 128, generator=g).to('cuda')
        with torch.no_grad():
            supp_activs = model(supp_imgs, mode='supp_bb')                                            # :343
            qry_feats = model(qry_imgs, mode='bb')
            qry_activs, qry_box_out = model(qry_feats, mode='not_cls')                                # :349
            qry_activs = [a.clone() for a in qry_activs]
            qry_box_out = [b.clone() for b in qry_box_out]
        # ---- inner step(s) on the support set
        anch_confs, obj_embds = model(supp_activs, fast_weights=None, mode='supp_cls')                # :563
        assert len(anch_confs) == 3 and anch_confs[0].requires_grad
        cls_logits = torch.cat([c.movedim(1, 3).reshape(-1) for c in anch_confs])
        target = torch.rand(cls_logits.shape, generator=g).to('cuda')
        supp_class_loss = F.binary_cross_entropy_with_logits(cls_logits, target)                      # :656
        inner_grad = torch.autograd.grad(supp_class_loss, model.class_net.parameters(), allow_unused=True, only_inputs=True,
                                         create_graph=True)                                           # :658
        fast_weights = []
        for p_ix, (n, par) in enumerate(model.class_net.named_parameters()):                          # :660-678
            if 'bn_' in n:
                update_par = par
            else:
                par_lr = learnable_lr[-2] if 'predict_dw' in n else (learnable_lr[-1] if 'predict_p' in n else learnable_lr[int(n[7])])
                assert inner_grad[p_ix] is not None, n
                update_par = par - par_lr * inner_grad[p_ix]
            fast_weights.append(update_par)
        # ---- query pass with the fast weights, loss, backward
        qry_class_out = model(qry_activs, fast_weights=fast_weights, mode='qry_cls')                  # :681
        sizes = [c.shape[-1] for c in qry_class_out]
        rs = np.random.RandomState(1)
        qry_cls_anchors = [torch.from_numpy(rs.choice([-2, -1, -1, -1, 0], size=(2, s, s, 9)).astype(np.int64)).to('cuda') for s in sizes]
        qry_bbox_anchors = [torch.from_numpy((rs.normal(0, 0.2, (2, s, s, 36)) * (rs.uniform(size=(2, s, s, 36)) < 0.3)).astype(np.float32)).to('cuda') for s in sizes]
        qry_num_positives = torch.tensor([5.0, 3.0], device='cuda')
        qry_loss, qry_class_loss, qry_box_loss = loss_fn(qry_class_out, qry_box_out, qry_cls_anchors, qry_bbox_anchors, qry_num_positives)  # :683
        qry_loss.backward()                                                                           # :687
        got = {n: p.grad for n, p in model.class_net.named_parameters()}
        assert all(v is not None and bool(torch.isfinite(v).all()) for v in got.values())
        assert float(got['predict_pw'].abs().max()) > 0 and float(got['conv_pw0'].abs().max()) > 0
        assert all(lr.grad is not None and bool(torch.isfinite(lr.grad).all()) for lr in learnable_lr)
        assert any(float(lr.grad.abs()) > 0 for lr in learnable_lr)
        # the inner step moved the query outputs
        with torch.no_grad():
            base = model(qry_activs, fast_weights=None, mode='qry_cls')
        assert not torch.equal(base[0], qry_class_out[0].detach())
    finally:
        sys.path.remove(os.path.join(root, 'ood_object_detection_amd'))
        for k in [k for k in sys.modules if k == 'effdet' or k.startswith('effdet.')]:
            del sys.modules[k]


def test_integration_md_ctypes_stubs_run():
    """the ctypes bindings printed in INTEGRATION.md (section B) are executed verbatim and compared with the package's own
    callables: documentation that cannot rot"""
    import re
    from ood_object_detection_amd import _lib as L
    from ood_object_detection_amd.effdet import bench as pb, soft_nms as ps
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, 'INTEGRATION.md')).read()
    sec = text[text.index('## B.'):text.index('### anchors.py:95')]
    blocks = re.findall(r'```python\n(.*?)```', sec, flags=re.S)
    assert len(blocks) >= 3
    L.load()
    ns = {}
    for b in blocks:
        exec(b.replace("ctypes.CDLL('libeffdet_hip.so')", "ctypes.CDLL(%r)" % L.LIB_PATH), ns)
    dev = 'cuda:0'
    g = torch.Generator().manual_seed(5)
    # _post_process
    B, C, A = 2, 7, 9
    sizes = [16, 8, 4, 2, 1]
    cls = [torch.randn(B, A * C, s, s, generator=g).to(dev) for s in sizes]
    box = [torch.randn(B, A * 4, s, s, generator=g).to(dev) for s in sizes]
    got = ns['_post_process'](cls, box, 5, C, 200)
    ref = pb._post_process(cls, box, 5, C, 200)
    for a, r in zip(got, ref):
        assert torch.equal(a, r)
    # batched_soft_nms
    n = 300
    xy = torch.rand(n, 2, generator=g) * 100
    boxes = torch.cat([xy, xy + 5 + torch.rand(n, 2, generator=g) * 40], 1).to(dev)
    scores = torch.rand(n, generator=g).to(dev)
    idxs = torch.randint(0, 4, (n,), generator=g).to(dev)
    k1, s1 = ns['batched_soft_nms'](boxes, scores, idxs, True, 0.5, 0.3, 0.001)
    k2, s2 = ps.batched_soft_nms(boxes, scores, idxs, method_gaussian=True, sigma=0.5, iou_threshold=0.3, score_threshold=0.001)
    m = min(len(k1), len(k2))
    assert m > 0 and torch.equal(k1[:m], k2[:m]) and torch.allclose(s1[:m], s2[:m], atol=1e-6)
