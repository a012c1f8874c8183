#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE's own Python modules.

Build-container only: needs /root/reference (read-only) and the import stubs under
tests/oracle_stubs/ (see the README there).  Nothing from the reference is copied; the
fixtures hold inputs and the reference's outputs only.

    PYTORCH_JIT=0 python tools/make_golden.py

Fixtures written
    anchors.npz               Anchors(...).boxes digests for 512/640/768/1024 (+ every 97th row)
    post_process.npz          bench._post_process on seeded logits
    decode.npz                anchors.decode_box_outputs / clip_boxes_xyxy
    soft_nms.npz              soft_nms.soft_nms / batched_soft_nms (gaussian + linear)
    generate_detections.npz   anchors.generate_detections (soft path: pure reference; hard path:
                              NMS step comes from the oracle via the torchvision stub)
    loss.npz                  loss.loss_fn values and autograd gradients
    evaluation.npz            ObjectDetectionEvaluator mAP / CorLoc on seeded detections
    labeler.npz               anchors.AnchorLabeler.batch_label_anchors
    config.npz                model_config.get_efficientdet_config + fpn_config.bifpn_config dumps
    bifpn_head.npz            EfficientDet(config) forward (reference BiFpn/HeadNet code on stub
                              conv layers and the oracle backbone): key/shape list + outputs
    bifpn_head_pad0.npz       the same for efficientdet_d0 / d1 (pad_type '', redundant_bias False)
    aux_losses.npz            loss.cosine_loss / smooth_l1_loss / l2_loss / SupportLoss values (+ one gradient each)
    meta_nets.npz             the reference's own MetaHead / AnchorNet / ProjectionNet classes
                              (efficientdet.py:569-830) on seeded weights and inputs: forwards, fast_weights,
                              level_offset, separate head, weighted_median, encoding tables
"""
import hashlib
import json
import os
import sys

os.environ.setdefault('PYTORCH_JIT', '0')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'tests', 'oracle_stubs'), '/root/reference', ROOT,
                os.path.join(ROOT, 'tests')]

import numpy as np
import torch

from _seeded import eval_case, seeded_array, seeded_tensor

OUT = os.path.join(ROOT, 'tests', 'golden')
torch.set_num_threads(4)


def save(name, **arrays):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print('%-28s %8.1f KB' % (name + '.npz', os.path.getsize(path) / 1024.))


def gen_anchors():
    from effdet.anchors import Anchors
    out = {}
    ratios = [(1.0, 1.0), (1.4, 0.7), (0.7, 1.4)]
    for s in (128, 512, 640, 768, 1024):
        b = Anchors(3, 7, 3, ratios, 4.0, (s, s)).boxes
        raw = b.numpy().tobytes()
        out['n_%d' % s] = np.int64(b.shape[0])
        out['sha256_%d' % s] = np.frombuffer(hashlib.sha256(raw).digest(), dtype=np.uint8)
        out['rows97_%d' % s] = b[::97].numpy()
        out['last_%d' % s] = b[-1].numpy()
    # non-square + scalar aspect ratios + per-level anchor scales
    b = Anchors(3, 6, 2, [1.0, 2.0, 0.5], [4.0, 3.0, 4.0, 5.0], (128, 256)).boxes
    out['odd_full'] = b.numpy()
    save('anchors', **out)


def gen_post_process():
    from effdet.bench import _post_process
    B, C, A, k = 2, 7, 9, 200
    sizes = [16, 8, 4, 2, 1]
    cls = [torch.from_numpy(seeded_array(1, 'cls%d' % i, (B, A * C, s, s), scale=2.0)) for i, s in enumerate(sizes)]
    box = [torch.from_numpy(seeded_array(1, 'box%d' % i, (B, A * 4, s, s), scale=0.5)) for i, s in enumerate(sizes)]
    c, b, idx, cl = _post_process(cls, box, 5, C, k)
    # torch.topk tie order is unspecified: make sure this fixture has no ties inside the top-k
    flat = torch.cat([x.permute(0, 2, 3, 1).reshape(B, -1) for x in cls], 1)
    for i in range(B):
        top = torch.sort(flat[i], descending=True)[0][:k + 1]
        assert (top[:-1] > top[1:]).all(), 'tie in fixture, change the seed'
    save('post_process', cls_topk=c, box_topk=b, indices=idx, classes=cl,
         meta=np.array([B, C, A, k] + sizes))


def gen_decode():
    from effdet.anchors import decode_box_outputs, clip_boxes_xyxy, Anchors
    anchors = Anchors(3, 7, 3, [(1.0, 1.0), (1.4, 0.7), (0.7, 1.4)], 4.0, (128, 128)).boxes
    n = 512
    rs = np.random.RandomState(5)
    sel = torch.from_numpy(rs.randint(0, anchors.shape[0], n))
    rel = torch.from_numpy(seeded_array(2, 'rel', (n, 4), scale=0.7))
    a = anchors[sel]
    yxyx = decode_box_outputs(rel, a, output_xyxy=False)
    xyxy = decode_box_outputs(rel, a, output_xyxy=True)
    clipped = clip_boxes_xyxy(xyxy, torch.tensor([100., 120.]))
    save('decode', rel=rel, anchors=a, yxyx=yxyx, xyxy=xyxy, clipped=clipped, clip_size=np.array([100., 120.], dtype=np.float32))


def _rand_boxes(seed, n, extent=200.):
    rs = np.random.RandomState(seed)
    # clustered boxes so that IoUs are often large
    centers = rs.uniform(20, extent - 20, (max(1, n // 6), 2))
    c = centers[rs.randint(0, centers.shape[0], n)] + rs.normal(0, 4.0, (n, 2))
    wh = rs.uniform(8, 60, (n, 2))
    b = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
    return torch.from_numpy(b)


def gen_soft_nms():
    from effdet.soft_nms import soft_nms, batched_soft_nms
    out = {}
    for tag, n, ncls in (('a', 300, 4), ('b', 57, 1), ('c', 1, 1)):
        boxes = _rand_boxes(10 + n, n)
        scores = torch.from_numpy(np.random.RandomState(20 + n).uniform(0.011, 0.99, n).astype(np.float32))
        classes = torch.from_numpy(np.random.RandomState(30 + n).randint(0, ncls, n).astype(np.int64))
        out[tag + '_boxes'], out[tag + '_scores'], out[tag + '_classes'] = boxes, scores, classes
        i, s = soft_nms(boxes, scores, method_gaussian=True, sigma=0.5, iou_threshold=0.3, score_threshold=0.001)
        out[tag + '_g_idx'], out[tag + '_g_scores'] = i, s
        i, s = soft_nms(boxes, scores, method_gaussian=False, sigma=0.5, iou_threshold=0.3, score_threshold=0.001)
        out[tag + '_l_idx'], out[tag + '_l_scores'] = i, s
        i, s = batched_soft_nms(boxes, scores, classes, method_gaussian=True, iou_threshold=0.3, score_threshold=0.001)
        out[tag + '_bg_idx'], out[tag + '_bg_scores'] = i, s
    i, s = batched_soft_nms(torch.zeros(0, 4), torch.zeros(0), torch.zeros(0, dtype=torch.int64))
    out['empty_idx'], out['empty_scores'] = i, s
    save('soft_nms', **out)


def gen_generate_detections():
    from effdet.anchors import generate_detections, Anchors
    from effdet.bench import _post_process
    anchors = Anchors(3, 7, 3, [(1.0, 1.0), (1.4, 0.7), (0.7, 1.4)], 4.0, (128, 128)).boxes
    B, C, A, k = 3, 5, 9, 400
    sizes = [16, 8, 4, 2, 1]
    # logits centred so that a fair share passes sigmoid > 0.01 and NMS has real work
    cls = [torch.from_numpy(seeded_array(3, 'cls%d' % i, (B, A * C, s, s), scale=2.5)) - 2.0 for i, s in enumerate(sizes)]
    box = [torch.from_numpy(seeded_array(3, 'box%d' % i, (B, A * 4, s, s), scale=0.4)) for i, s in enumerate(sizes)]
    c, b, idx, cl = _post_process(cls, box, 5, C, k)
    out = dict(cls_topk=c, box_topk=b, indices=idx, classes=cl, anchors=anchors,
               meta=np.array([B, C, A, k] + sizes))
    scale = torch.tensor([1.0, 1.7, 0.6])
    size = torch.tensor([[128., 128.], [200., 160.], [70., 75.]])
    for i in range(B):
        for soft in (False, True):
            tag = 'soft' if soft else 'hard'
            out['det_%s_%d' % (tag, i)] = generate_detections(
                c[i], b[i], anchors, idx[i], cl[i], None, torch.tensor(128), max_det_per_image=100, soft_nms=soft)
            out['det_%s_info_%d' % (tag, i)] = generate_detections(
                c[i], b[i], anchors, idx[i], cl[i], scale[i], size[i], max_det_per_image=20, soft_nms=soft)
    out['img_scale'], out['img_size'] = scale, size
    save('generate_detections', **out)


def gen_loss():
    from effdet.loss import loss_fn
    B, C, A = 2, 6, 9
    sizes = [8, 4, 2, 1, 1]
    rs = np.random.RandomState(7)
    cls_out = [torch.from_numpy(seeded_array(4, 'c%d' % i, (B, A * C, s, s), scale=1.5)).requires_grad_() for i, s in enumerate(sizes)]
    box_out = [torch.from_numpy(seeded_array(4, 'b%d' % i, (B, A * 4, s, s), scale=0.3)).requires_grad_() for i, s in enumerate(sizes)]
    cls_t = [torch.from_numpy(rs.choice([-2, -1, -1, -1, 0, 1, 2, 3, 4, 5], size=(B, s, s, A)).astype(np.int64)) for s in sizes]
    box_t = []
    for i, s in enumerate(sizes):
        t = seeded_array(4, 'bt%d' % i, (B, s, s, A * 4), scale=0.2)
        t[rs.uniform(size=t.shape) < 0.6] = 0.0
        box_t.append(torch.from_numpy(t))
    npos = torch.tensor([5., 3.])
    out = {}
    for tag, alpha, w, ls in (('pre', 0.15, 50.0, 0.0), ('inf', 0.25, 5.0, 0.0), ('ls', 0.25, 5.0, 0.1)):
        total, cl, bl = loss_fn(cls_out, box_out, cls_t, box_t, npos, num_classes=C, alpha=alpha, gamma=1.5,
                                delta=0.1, box_loss_weight=w, label_smoothing=ls)
        grads = torch.autograd.grad(total, cls_out + box_out)
        out[tag + '_loss'] = torch.stack([total, cl, bl]).detach()
        for i in range(5):
            out['%s_gc%d' % (tag, i)] = grads[i]
            out['%s_gb%d' % (tag, i)] = grads[5 + i]
    for i in range(5):
        out['cls_t%d' % i], out['box_t%d' % i] = cls_t[i], box_t[i]
    out['npos'] = npos
    out['meta'] = np.array([B, C, A] + sizes)
    save('loss', **out)


def gen_labeler():
    from effdet.anchors import Anchors, AnchorLabeler
    anchors = Anchors(3, 7, 3, [(1.0, 1.0), (1.4, 0.7), (0.7, 1.4)], 4.0, (128, 128))
    lab = AnchorLabeler(anchors, num_classes=6, match_threshold=0.5)
    rs = np.random.RandomState(11)
    gt_boxes, gt_cls = [], []
    for n in (4, 1, 0, 7):
        y0 = rs.uniform(0, 90, n); x0 = rs.uniform(0, 90, n)
        h = rs.uniform(6, 60, n); w = rs.uniform(6, 60, n)
        gt_boxes.append(torch.from_numpy(np.stack([y0, x0, np.minimum(y0 + h, 128), np.minimum(x0 + w, 128)], 1).astype(np.float32)).reshape(n, 4))
        c = rs.randint(1, 7, n).astype(np.int64)
        if n == 7:
            c[2] = -1   # filtered by filter_valid
        gt_cls.append(torch.from_numpy(c))
    # AnchorLabeler.batch_label_anchors itself raises under torch 2.10 (anchors.py:428 `.view` on the
    # transposed box-coder output - an ordinary RuntimeError), so the fixture pins what it calls:
    # TargetAssigner.assign per image (anchors.py:415) plus its `cls - 1` and num_positives rules
    # (:418, :436).  The per-level unpack (:421-433) is a plain reshape of these flat arrays.
    from effdet.object_detection import BoxList
    out = {}
    npos = []
    for i in range(4):
        valid = gt_cls[i] > -1
        c, b, m = lab.target_assigner.assign(BoxList(anchors.boxes), BoxList(gt_boxes[i][valid]), gt_cls[i][valid])
        out['cls_flat%d' % i] = (c - 1).long()
        out['box_flat%d' % i] = b.contiguous()
        out['match%d' % i] = m.match_results
        npos.append((m.match_results > -1).float().sum())
        out['gt_boxes%d' % i], out['gt_cls%d' % i] = gt_boxes[i], gt_cls[i]
    out['npos'] = torch.stack(npos)
    save('labeler', **out)


def gen_config():
    from effdet.config import get_efficientdet_config, get_fpn_config
    out = {}
    for name in ('tf_efficientdet_d0', 'tf_efficientdet_d1', 'tf_efficientdet_d2', 'tf_efficientdet_d3',
                 'tf_efficientdet_d4', 'tf_efficientdet_d5'):
        h = get_efficientdet_config(name)
        out[name] = np.array(json.dumps({k: h[k] for k in sorted(h.keys())}, default=list))
    nodes = get_fpn_config('bifpn_fa', 3, 7).nodes
    out['bifpn_fa_3_7'] = np.array(json.dumps([dict(n) for n in nodes], default=list))
    nodes = get_fpn_config('bifpn_sum', 3, 8).nodes
    out['bifpn_sum_3_8'] = np.array(json.dumps([dict(n) for n in nodes], default=list))
    save('config', **out)


def gen_bifpn_head():
    """EfficientDet(config).forward through the reference's BiFpn / HeadNet / FpnCombine code."""
    from absl import flags
    from effdet.config import get_efficientdet_config
    from effdet.efficientdet import EfficientDet
    out = {}
    for tag, name, size, ncls, seed in (('d0', 'tf_efficientdet_d0', 128, 3, 21), ('d1', 'tf_efficientdet_d1', 128, 2, 22)):
        flags.FLAGS.pretrain_classes = ncls
        cfg = get_efficientdet_config(name)
        cfg.image_size = (size, size)
        model = EfficientDet(cfg, pretrained_backbone=False).eval()
        sd = model.state_dict()
        keys = [k for k in sd.keys() if not k.endswith('num_batches_tracked')]
        new = {k: seeded_tensor(seed, k, sd[k].shape) for k in keys}
        for k in sd.keys():
            if k.endswith('num_batches_tracked'):
                new[k] = sd[k]
        model.load_state_dict(new, strict=True)
        x = torch.from_numpy(seeded_array(seed, 'input', (2, 3, size, size)))
        with torch.no_grad():
            cls_o, box_o = model(x)
            feats, activs = model(x, mode='fpn')
        out[tag + '_keys'] = np.array(keys)
        out[tag + '_shapes'] = np.array([json.dumps(list(sd[k].shape)) for k in keys])
        out[tag + '_meta'] = np.array([size, ncls, seed])
        for i in range(5):
            out['%s_cls%d' % (tag, i)] = cls_o[i]
            out['%s_box%d' % (tag, i)] = box_o[i]
            out['%s_act%d' % (tag, i)] = activs[i]
    flags.FLAGS.pretrain_classes = 400
    save('bifpn_head', **out)


def gen_bifpn_head_pad0():
    """The PyTorch-trained model family (efficientdet_d0 / d1 on efficientnet_b0 / b1, the scripts' default models,
    pretrain.py:81-112, infer.py:119-149): pad_type = '' (static symmetric padding - the variant the timm-layer stub reproduces
    EXACTLY: nn.Conv2d / nn.MaxPool2d with padding=((s-1)+(k-1))//2) and redundant_bias = False (bias-less separable convs).
    EfficientDet(config).forward through the reference's BiFpn / HeadNet / FpnCombine code on seeded weights."""
    from absl import flags
    from effdet.config import get_efficientdet_config
    from effdet.efficientdet import EfficientDet
    out = {}
    for tag, name, size, ncls, seed in (('d0', 'efficientdet_d0', 128, 3, 31), ('d1', 'efficientdet_d1', 128, 2, 32)):
        flags.FLAGS.pretrain_classes = ncls
        cfg = get_efficientdet_config(name)
        assert cfg.pad_type == '' and not cfg.redundant_bias
        cfg.image_size = (size, size)
        model = EfficientDet(cfg, pretrained_backbone=False).eval()
        sd = model.state_dict()
        keys = [k for k in sd.keys() if not k.endswith('num_batches_tracked')]
        new = {k: seeded_tensor(seed, k, sd[k].shape) for k in keys}
        for k in sd.keys():
            if k.endswith('num_batches_tracked'):
                new[k] = sd[k]
        model.load_state_dict(new, strict=True)
        x = torch.from_numpy(seeded_array(seed, 'input', (2, 3, size, size)))
        with torch.no_grad():
            cls_o, box_o = model(x)
            feats, activs = model(x, mode='fpn')
        out[tag + '_keys'] = np.array(keys)
        out[tag + '_shapes'] = np.array([json.dumps(list(sd[k].shape)) for k in keys])
        out[tag + '_meta'] = np.array([size, ncls, seed])
        for i in range(5):
            out['%s_cls%d' % (tag, i)] = cls_o[i]
            out['%s_box%d' % (tag, i)] = box_o[i]
            out['%s_act%d' % (tag, i)] = activs[i]
    flags.FLAGS.pretrain_classes = 400
    save('bifpn_head_pad0', **out)


def gen_meta_nets():
    """The reference's MetaHead / AnchorNet / ProjectionNet (effdet/efficientdet.py:569-830), instantiated on the CPU.
    Their constructors read absl FLAGS (supplied by the stub namespace below) and move a few buffers with
    `.to('cuda')` / `.cuda()`; in THIS script only, those two calls are mapped to the CPU so the classes can be built in
    the GPU-less container.  Everything else - constructor, parameter order, forward - is the reference's code."""
    from absl import flags
    FL = flags.FLAGS
    from effdet.config import get_efficientdet_config
    import effdet.efficientdet as ref
    _to, _cuda = torch.Tensor.to, torch.Tensor.cuda

    def to_cpu(self, *a, **k):
        a = tuple('cpu' if (isinstance(v, str) and v.startswith('cuda')) else v for v in a)
        if isinstance(k.get('device'), str) and k['device'].startswith('cuda'):
            k['device'] = 'cpu'
        return _to(self, *a, **k)
    torch.Tensor.to = to_cpu
    torch.Tensor.cuda = lambda self, *a, **k: self
    out = {}
    try:
        cfg = get_efficientdet_config('tf_efficientdet_d0')
        Fc, A, L, R = cfg.fpn_channels, 9, cfg.num_levels, cfg.box_class_repeats
        sizes = [16, 8, 4, 2, 1]
        B = 3
        seed = 51
        # ---- MetaHead: pretrain_init = a seeded class_net state dict (infer.py:186-191)
        keys = {}
        for l in range(R):
            keys['class_net.conv_rep.%d.conv_dw.weight' % l] = (Fc, 1, 3, 3)
            keys['class_net.conv_rep.%d.conv_pw.weight' % l] = (Fc, Fc, 1, 1)
            keys['class_net.conv_rep.%d.conv_pw.bias' % l] = (Fc,)
            for lev in range(L):
                keys['class_net.bn_rep.%d.%d.bn.weight' % (l, lev)] = (Fc,)
                keys['class_net.bn_rep.%d.%d.bn.bias' % (l, lev)] = (Fc,)
        keys['class_net.predict.conv_dw.weight'] = (Fc, 1, 3, 3)
        init = {k: seeded_tensor(seed, k, shp) for k, shp in keys.items()}
        for sep in (False, True):
            FL.separate_head = sep
            mh = ref.MetaHead(cfg, pretrain_init=init)
            with torch.no_grad():
                mh.predict_pw.copy_(seeded_tensor(seed, 'meta.predict_pw', mh.predict_pw.shape) * (1.0 / Fc) ** 0.5)
                mh.predict_pb.copy_(seeded_tensor(seed, 'meta.predict_pb', mh.predict_pb.shape))
                if sep:
                    mh.add_head()
                    mh.predict_pw_sep.data.copy_(seeded_tensor(seed, 'meta.predict_pw_sep', mh.predict_pw_sep.shape) * (1.0 / Fc) ** 0.5)
                    mh.predict_pb_sep.data.copy_(seeded_tensor(seed, 'meta.predict_pb_sep', mh.predict_pb_sep.shape))
            x = [torch.from_numpy(seeded_array(seed, 'lvl%d' % i, (B, Fc, s, s))) for i, s in enumerate(sizes)]
            with torch.no_grad():
                if not sep:
                    o, a = mh([t.clone() for t in x], ret_activs=True)
                    for i in range(L):
                        out['mh_out%d' % i], out['mh_act%d' % i] = o[i], a[i]
                    o2 = mh([t.clone() for t in x], level_offset=2)
                    assert len(o2) == L - 2
                    for i in range(L - 2):
                        out['mh_off2_out%d' % i] = o2[i]
                    # fast weights in the reference's list order (efficientdet.py:645-652): a perturbed copy of the parameters
                    fw = list(mh.conv_dw_rep) + list(mh.conv_pw_rep) + list(mh.conv_pb_rep) + list(mh.predict) + list(mh.bn_rep_w) + list(mh.bn_rep_b)
                    fw = [w.detach() + 0.05 * seeded_tensor(seed, 'fw%d' % i, w.shape) for i, w in enumerate(fw)]
                    o3, a3 = mh([t.clone() for t in x], fast_weights=fw, ret_activs=True)
                    for i in range(L):
                        out['mh_fw_out%d' % i], out['mh_fw_act%d' % i] = o3[i], a3[i]
                    out['mh_n_fast'] = np.int64(len(fw))
                    # the order of named parameters is part of the contract (optimizers / checkpoints of infer.py)
                    out['mh_param_names'] = np.array([n for n, _ in mh.named_parameters()])
                else:
                    co, ao, act = mh([t.clone() for t in x], ret_activs=True, level_offset=FL.supp_level_offset, heads='both')
                    out['mh_both_levels'] = np.int64(len(co))
                    for i in range(len(co)):
                        out['mh_both_cls%d' % i], out['mh_both_anch%d' % i], out['mh_both_act%d' % i] = co[i], ao[i], act[i]
        out['mh_meta'] = np.array([seed, B, Fc, A, L, R] + sizes)
        out['supp_level_offset_default'] = np.int64(2)            # infer.py:94 / pretrain.py:63
        # ---- AnchorNet (FLAGS.num_anch_layers 3 and 1; eval-mode BN with seeded running statistics)
        FL.supp_alpha, FL.learn_alpha, FL.inner_alpha, FL.detach_anch = False, False, 0.25, False
        for layers in (3, 1):
            FL.num_anch_layers = layers
            net = ref.AnchorNet(cfg).eval()
            sd = net.state_dict()
            new = {k: (v if k.endswith('num_batches_tracked') else seeded_tensor(seed + layers, k, v.shape)) for k, v in sd.items()}
            net.load_state_dict(new, strict=True)
            x = [torch.from_numpy(seeded_array(seed + 1, 'an%d' % i, (2, Fc, s, s))) for i, s in enumerate(sizes)]
            with torch.no_grad():
                o = net([t.clone() for t in x])
            out['an%d_keys' % layers] = np.array([k for k in sd.keys() if not k.endswith('num_batches_tracked')])
            out['an%d_shapes' % layers] = np.array([json.dumps(list(sd[k].shape)) for k in sd.keys() if not k.endswith('num_batches_tracked')])
            out['an%d_eps' % layers] = np.float64(net.bn_rep[0][0].bn.eps if layers > 1 else 0.0)
            for i in range(L):
                out['an%d_out%d' % (layers, i)] = o[i]
        # ---- ProjectionNet: encoding tables, MLP for every proj_depth, weighted_median
        FL.dot_mult, FL.dot_add, FL.median_grad = 3.0, 3.0, False
        for depth in (2, 3, 4):
            FL.proj_depth = depth
            net = ref.ProjectionNet(cfg, 128)
            lin = [m for m in net.projection if isinstance(m, torch.nn.Linear)]
            with torch.no_grad():
                for i, m in enumerate(lin):
                    m.weight.copy_(seeded_tensor(seed + depth, 'proj%d' % i, m.weight.shape) * (1.0 / m.in_features) ** 0.5)
                xin = torch.from_numpy(seeded_array(seed + 2, 'px', (5, 37, Fc + 42)))
                out['pn%d_out' % depth] = net(xin)
            out['pn%d_dims' % depth] = np.array([m.weight.shape[0] for m in lin] + [lin[0].weight.shape[1]])
        out['pn_anch_enc'], out['pn_cell_enc'], out['pn_lev_enc'] = net.anch_enc, net.cell_enc, net.lev_enc
        out['pn_dot'] = np.array([float(net.dot_mult), float(net.dot_add)])
        g = torch.Generator().manual_seed(2)
        for n in (1, 7, 200, 1024):
            e = torch.randn(n, 24, generator=g)
            c = torch.rand(n, generator=g)
            med, cs = net.weighted_median(e, c)
            out['wm%d_e' % n], out['wm%d_c' % n], out['wm%d_med' % n], out['wm%d_sum' % n] = e, c, med, cs
    finally:
        torch.Tensor.to, torch.Tensor.cuda = _to, _cuda
    save('meta_nets', **out)


def gen_aux_losses():
    """the episode-level losses infer.py / pretrain.py import next to DetectionLoss (effdet/loss.py:97-168, 404-439)"""
    from absl import flags
    from effdet.config import get_efficientdet_config
    from effdet.loss import SupportLoss, cosine_loss, l2_loss, smooth_l1_loss
    out = {}
    x = torch.from_numpy(seeded_array(71, 'x', (400,), scale=0.8))
    t = torch.from_numpy((seeded_array(71, 't', (400,)) > 0.3).astype(np.float32) * 2 - 1)
    w = torch.from_numpy(seeded_array(71, 'w', (400,), kind='uniform'))
    tgt = torch.from_numpy(seeded_array(71, 'tgt', (400,), scale=0.5))
    out['cos_0'] = cosine_loss(x.clone(), t, margin=0.)
    out['cos_m'] = cosine_loss(x.clone(), t, margin=0.2)
    a, b, c = smooth_l1_loss(x.clone(), tgt, beta=1. / 9, weights=w)
    out['sl1_b9'] = torch.stack([a, b, c])
    out['sl1_b9_mean'] = smooth_l1_loss(x.clone(), tgt, beta=1. / 9, weights=w.clone(), size_average=True)
    out['sl1_b0_mean'] = smooth_l1_loss(x.clone(), tgt, beta=0.0, size_average=True)     # (beta < 1e-5 with weights raises in the reference)
    a, b, c = l2_loss(x.clone(), tgt, weights=w)
    out['l2'] = torch.stack([a, b, c])
    cfg = get_efficientdet_config('tf_efficientdet_d0')
    cfg.num_classes = 1
    sizes = [8, 4, 2]
    co = [torch.from_numpy(seeded_array(72, 'co%d' % i, (2, 9, s, s), scale=1.5)) for i, s in enumerate(sizes)]
    ct = [torch.from_numpy(seeded_array(72, 'ct%d' % i, (2, 9, s, s), kind='uniform')) for i, s in enumerate(sizes)]
    npos = torch.tensor([3., 5.])
    for lt in ('ce', 'mse'):
        for tag, alpha, ls in (('a25', 0.25, 0.0), ('none', None, 0.0), ('ls', 0.25, 0.1)):
            cfg.label_smoothing = ls
            leaf = [c.clone().requires_grad_() for c in co]
            v = SupportLoss(cfg, lt)(leaf, ct, npos, alpha)
            out['sup_%s_%s' % (lt, tag)] = v.detach()
            out['sup_%s_%s_g0' % (lt, tag)] = torch.autograd.grad(v, leaf)[0]
    out['meta'] = np.array(sizes)
    save('aux_losses', **out)


def gen_evaluation():
    """mAP / CorLoc of the reference's ObjectDetectionEvaluator (effdet/evaluation/detection_evaluator.py:96-316) the way
    pretrain.py:246-252 drives it."""
    np.float, np.bool, np.NAN = float, bool, np.nan      # aliases numpy >= 1.24 / 2.0 dropped; the reference predates that
    from effdet.evaluation.detection_evaluator import ObjectDetectionEvaluator
    out = {}
    for tag, seed, n_img, C, n_det in (('a', 31, 10, 6, 40), ('b', 32, 3, 20, 100), ('c', 33, 1, 4, 12)):
        cats = [{'id': i + 1, 'name': 'c%d' % i} for i in range(C)]
        ev = ObjectDetectionEvaluator(cats, evaluate_corlocs=True)
        for i, im in enumerate(eval_case(seed, n_img, C, n_det)):
            ev.add_single_ground_truth_image_info(i, {'bbox': im['gt_boxes'], 'cls': im['gt_classes']})
            ev.add_single_detected_image_info(i, {'bbox': im['det_boxes'], 'scores': im['det_scores'], 'cls': im['det_classes']})
        with np.errstate(all='ignore'):
            m = ev.evaluate(['c%d' % i for i in range(C)])
        out[tag + '_meta'] = np.array([seed, n_img, C, n_det])
        out[tag + '_map'] = np.array(m['Precision/mAP@0.5IOU'])
        out[tag + '_corloc'] = np.array(m['Precision/meanCorLoc@0.5IOU'])
        out[tag + '_ap'] = np.array([m['AP@0.5IOU/c%d' % i] for i in range(C)])
        out[tag + '_cl'] = np.array([m['CorLoc@0.5IOU/c%d' % i] for i in range(C)])
    save('evaluation', **out)


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ['anchors', 'post_process', 'decode', 'soft_nms', 'generate_detections', 'loss',
                             'labeler', 'config', 'bifpn_head', 'bifpn_head_pad0', 'evaluation', 'meta_nets', 'aux_losses']
    for w in which:
        globals()['gen_' + w]()
