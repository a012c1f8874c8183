"""The oracle (oracle/*.py) against the fixtures the reference itself produced
(tools/make_golden.py).  CPU only."""
import hashlib
import json

import numpy as np
import pytest
import torch

from _seeded import seeded_array, seeded_state_dict
from oracle import model as om
from oracle import postprocess as op

RATIOS = [(1.0, 1.0), (1.4, 0.7), (0.7, 1.4)]


@pytest.mark.parametrize('size', [128, 512, 640, 768, 1024])
def test_anchors_bit_exact(golden, size):
    g = golden('anchors')
    b = op.anchor_boxes(3, 7, 3, RATIOS, 4.0, (size, size))
    assert b.shape[0] == int(g['n_%d' % size])
    assert hashlib.sha256(b.numpy().tobytes()).digest() == g['sha256_%d' % size].tobytes()
    assert np.array_equal(b[::97].numpy(), g['rows97_%d' % size])
    assert np.array_equal(b[0].numpy(), np.array([-12, -12, 20, 20], dtype=np.float32))


def test_anchor_counts_survey():
    # N for 512/640/768/1024
    for s, n in ((512, 49104), (640, 76725), (768, 110484), (1024, 196416)):
        assert op.anchor_boxes(3, 7, 3, RATIOS, 4.0, (s, s)).shape[0] == n


def test_anchors_odd_config(golden):
    g = golden('anchors')
    b = op.anchor_boxes(3, 6, 2, [1.0, 2.0, 0.5], [4.0, 3.0, 4.0, 5.0], (128, 256))
    assert np.array_equal(b.numpy(), g['odd_full'])


def _pp_inputs(seed, B, C, A, sizes, cs, bs, shift=0.0):
    cls = [torch.from_numpy(seeded_array(seed, 'cls%d' % i, (B, A * C, s, s), scale=cs)) - shift for i, s in enumerate(sizes)]
    box = [torch.from_numpy(seeded_array(seed, 'box%d' % i, (B, A * 4, s, s), scale=bs)) for i, s in enumerate(sizes)]
    return cls, box


def test_post_process(golden):
    g = golden('post_process')
    B, C, A, k = [int(v) for v in g['meta'][:4]]
    sizes = [int(v) for v in g['meta'][4:]]
    cls, box = _pp_inputs(1, B, C, A, sizes, 2.0, 0.5)
    c, b, idx, cl = op.post_process(cls, box, 5, C, k)
    assert np.array_equal(idx.numpy(), g['indices'])
    assert np.array_equal(cl.numpy(), g['classes'])
    assert np.array_equal(c.numpy(), g['cls_topk'])
    assert np.array_equal(b.numpy(), g['box_topk'])


def test_decode_and_clip(golden):
    g = golden('decode')
    rel, a = torch.from_numpy(g['rel']), torch.from_numpy(g['anchors'])
    assert np.array_equal(op.decode_box_outputs(rel, a, False).numpy(), g['yxyx'])
    xyxy = op.decode_box_outputs(rel, a, True)
    assert np.array_equal(xyxy.numpy(), g['xyxy'])
    assert np.array_equal(op.clip_boxes_xyxy(xyxy, torch.from_numpy(g['clip_size'])).numpy(), g['clipped'])


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_soft_nms(golden, tag):
    g = golden('soft_nms')
    boxes, scores, classes = (torch.from_numpy(g[tag + s]) for s in ('_boxes', '_scores', '_classes'))
    i, s = op.soft_nms(boxes, scores, True, 0.5, 0.3, 0.001)
    assert np.array_equal(i.numpy(), g[tag + '_g_idx']) and np.array_equal(s.numpy(), g[tag + '_g_scores'])
    i, s = op.soft_nms(boxes, scores, False, 0.5, 0.3, 0.001)
    assert np.array_equal(i.numpy(), g[tag + '_l_idx']) and np.array_equal(s.numpy(), g[tag + '_l_scores'])
    i, s = op.batched_soft_nms(boxes, scores, classes, True, 0.5, 0.3, 0.001)
    assert np.array_equal(i.numpy(), g[tag + '_bg_idx']) and np.array_equal(s.numpy(), g[tag + '_bg_scores'])
    # early stop == prefix of the exhaustive run
    i2, s2 = op.batched_soft_nms(boxes, scores, classes, True, 0.5, 0.3, 0.001, max_picks=10)
    assert np.array_equal(i2.numpy(), g[tag + '_bg_idx'][:10]) and np.array_equal(s2.numpy(), g[tag + '_bg_scores'][:10])


def test_soft_nms_empty(golden):
    g = golden('soft_nms')
    i, s = op.batched_soft_nms(torch.zeros(0, 4), torch.zeros(0), torch.zeros(0, dtype=torch.int64))
    assert i.shape == g['empty_idx'].shape and s.shape == g['empty_scores'].shape


@pytest.mark.parametrize('soft', [False, True])
def test_generate_detections(golden, soft):
    g = golden('generate_detections')
    B = int(g['meta'][0])
    anchors = torch.from_numpy(g['anchors'])
    tag = 'soft' if soft else 'hard'
    for i in range(B):
        args = [torch.from_numpy(g[k][i]) for k in ('cls_topk', 'box_topk')] + [anchors] + \
               [torch.from_numpy(g[k][i]) for k in ('indices', 'classes')]
        det = op.generate_detections(*args, None, torch.tensor(128), 100, soft)
        assert np.array_equal(det.numpy(), g['det_%s_%d' % (tag, i)])
        det = op.generate_detections(*args, torch.from_numpy(g['img_scale'])[i], torch.from_numpy(g['img_size'])[i], 20, soft)
        assert np.array_equal(det.numpy(), g['det_%s_info_%d' % (tag, i)])


@pytest.mark.parametrize('tag', ['d0', 'd1'])
def test_bifpn_head_wiring(golden, tag):
    """oracle forward == the reference's BiFpn/HeadNet/EfficientDet.forward on the same weights."""
    from ood_object_detection_amd.effdet.config import get_efficientdet_config, get_fpn_config
    g = golden('bifpn_head')
    size, ncls, seed = [int(v) for v in g[tag + '_meta']]
    keys = [str(k) for k in g[tag + '_keys']]
    shapes = [json.loads(str(s)) for s in g[tag + '_shapes']]
    sd = seeded_state_dict(seed, keys, shapes)
    cfg = get_efficientdet_config({'d0': 'tf_efficientdet_d0', 'd1': 'tf_efficientdet_d1'}[tag])
    cfg.image_size = (size, size)
    cfg.num_classes = ncls
    nodes = get_fpn_config(cfg.fpn_name, cfg.min_level, cfg.max_level).nodes
    x = torch.from_numpy(seeded_array(seed, 'input', (2, 3, size, size)))
    with torch.no_grad():
        cls_o, box_o = om.efficientdet_forward(sd, cfg, x, nodes)
        _, activs = om.efficientdet_forward(sd, cfg, x, nodes, mode='fpn')
    for i in range(5):
        assert np.array_equal(activs[i].numpy(), g['%s_act%d' % (tag, i)])
        assert np.array_equal(cls_o[i].numpy(), g['%s_cls%d' % (tag, i)])
        assert np.array_equal(box_o[i].numpy(), g['%s_box%d' % (tag, i)])


@pytest.mark.parametrize('tag', ['d0', 'd1'])
def test_bifpn_head_wiring_pad0(golden, tag):
    """efficientdet_d0 / d1 on efficientnet_b0 / b1 - pad_type '' (static symmetric padding: the variant the reference classes pin
    EXACTLY) and redundant_bias False, the default models of pretrain.py:81-112 / infer.py:119-149: oracle forward ==
    the reference's EfficientDet.forward on the same weights, bit for bit."""
    from ood_object_detection_amd.effdet.config import get_efficientdet_config, get_fpn_config
    g = golden('bifpn_head_pad0')
    size, ncls, seed = [int(v) for v in g[tag + '_meta']]
    keys = [str(k) for k in g[tag + '_keys']]
    shapes = [json.loads(str(s)) for s in g[tag + '_shapes']]
    sd = seeded_state_dict(seed, keys, shapes)
    cfg = get_efficientdet_config({'d0': 'efficientdet_d0', 'd1': 'efficientdet_d1'}[tag])
    assert cfg.pad_type == '' and cfg.redundant_bias is False
    cfg.image_size = (size, size)
    cfg.num_classes = ncls
    nodes = get_fpn_config(cfg.fpn_name, cfg.min_level, cfg.max_level).nodes
    x = torch.from_numpy(seeded_array(seed, 'input', (2, 3, size, size)))
    with torch.no_grad():
        cls_o, box_o = om.efficientdet_forward(sd, cfg, x, nodes)
        _, activs = om.efficientdet_forward(sd, cfg, x, nodes, mode='fpn')
    for i in range(5):
        assert np.array_equal(activs[i].numpy(), g['%s_act%d' % (tag, i)])
        assert np.array_equal(cls_o[i].numpy(), g['%s_cls%d' % (tag, i)])
        assert np.array_equal(box_o[i].numpy(), g['%s_box%d' % (tag, i)])


def test_normalize_matches_loader_expression():
    """oracle/preprocess.py vs the literal PrefetchLoader expression (effdet/data/loader.py:114-115,127-128)."""
    import torch
    from oracle import preprocess as opre
    g = torch.Generator().manual_seed(3)
    x = torch.randint(0, 256, (2, 3, 16, 24), generator=g, dtype=torch.uint8)
    mean = torch.tensor([v * 255 for v in opre.IMAGENET_DEFAULT_MEAN]).view(1, 3, 1, 1)
    std = torch.tensor([v * 255 for v in opre.IMAGENET_DEFAULT_STD]).view(1, 3, 1, 1)
    ref = x.float().sub_(mean).div_(std)
    got = opre.normalize_u8(x.numpy())
    assert np.array_equal(got, ref.numpy())


def test_auroc_oracle_matches_sklearn():
    sk = pytest.importorskip('sklearn.metrics')
    from oracle import postprocess as opp
    rng = np.random.RandomState(0)
    pos = np.round(rng.randn(200) + 0.7, 1)          # rounding plants ties
    neg = np.round(rng.randn(150), 1)
    ref = sk.roc_auc_score(np.r_[np.ones_like(pos), np.zeros_like(neg)], np.r_[pos, neg])
    assert abs(opp.auroc(pos, neg) - ref) < 1e-12


def test_clip_adam_oracle_matches_torch():
    """oracle/train.py vs torch.nn.utils.clip_grad_norm_ + torch.optim.Adam (what pretrain.py:272-276 calls)."""
    import torch
    from oracle import train as otr
    g0 = torch.Generator().manual_seed(4)
    shapes = [(7, 5), (33,), (4, 3, 3, 3)]
    params = [torch.nn.Parameter(torch.randn(*s, generator=g0)) for s in shapes]
    opt = torch.optim.Adam(params, lr=1e-3)
    p = np.concatenate([q.detach().numpy().ravel() for q in params]); m = np.zeros_like(p); v = np.zeros_like(p)
    for step in range(1, 4):
        grads = [torch.randn(*s, generator=g0) * (30.0 if step == 2 else 0.1) for s in shapes]    # step 2 clips
        for q, g in zip(params, grads):
            q.grad = g.clone()
        tn = torch.nn.utils.clip_grad_norm_(params, 10.)
        opt.step()
        p, m, v, n = otr.clip_adam_step(p, np.concatenate([g.numpy().ravel() for g in grads]), m, v, step)
        ref = np.concatenate([q.detach().numpy().ravel() for q in params])
        assert abs(float(tn) - float(n)) < 1e-4 * max(1.0, float(tn))
        assert np.abs(p - ref).max() < 2e-6


@pytest.mark.parametrize('hw', [(480, 640), (333, 500), (64, 48), (100, 37), (640, 640), (720, 1280)])
def test_resize_pad_matches_pil(hw):
    """oracle/preprocess.py::resize_pad vs the reference's ResizePad arithmetic executed with PIL itself
    (effdet/data/transforms.py:82-94): Image.new + resize(BILINEAR) + paste - bit exact, down- and up-scaling."""
    Image = pytest.importorskip('PIL.Image')
    from oracle import preprocess as opre
    rng = np.random.RandomState(hw[0] + hw[1])
    img = rng.randint(0, 256, (hw[0], hw[1], 3)).astype(np.uint8)
    target = 128
    fill = opre.resolve_fill_color('mean')
    pim = Image.fromarray(img)
    width, height = pim.size
    s = min(target / height, target / width)
    new_img = Image.new('RGB', (target, target), color=fill)
    new_img.paste(pim.resize((int(width * s), int(height * s)), Image.BILINEAR))
    got, inv_scale = opre.resize_pad(img, target, fill)
    assert np.array_equal(got, np.asarray(new_img))
    assert inv_scale == 1.0 / s


@pytest.mark.parametrize('tag,alpha,w,ls', [('pre', 0.15, 50.0, 0.0), ('inf', 0.25, 5.0, 0.0), ('ls', 0.25, 5.0, 0.1)])
def test_detection_loss_oracle_matches_reference(golden, tag, alpha, w, ls):
    """oracle/train.py::detection_loss (values + autograd gradients) vs the reference's loss_fn (effdet/loss.py:224-298)"""
    from oracle import train as ot
    g = golden('loss')
    B, C, A = [int(v) for v in g['meta'][:3]]
    sizes = [int(v) for v in g['meta'][3:]]
    cls_out = [torch.from_numpy(seeded_array(4, 'c%d' % i, (B, A * C, s, s), scale=1.5)).requires_grad_() for i, s in enumerate(sizes)]
    box_out = [torch.from_numpy(seeded_array(4, 'b%d' % i, (B, A * 4, s, s), scale=0.3)).requires_grad_() for i, s in enumerate(sizes)]
    cls_t = [torch.from_numpy(g['cls_t%d' % i]) for i in range(5)]
    box_t = [torch.from_numpy(g['box_t%d' % i]) for i in range(5)]
    total, cl, bl = ot.detection_loss(cls_out, box_out, cls_t, box_t, torch.from_numpy(g['npos']), C, alpha, 0.1, w, ls)
    got = torch.stack([total, cl, bl]).detach().numpy()
    assert np.allclose(got, g[tag + '_loss'], rtol=1e-6, atol=1e-7), (got, g[tag + '_loss'])
    grads = torch.autograd.grad(total, cls_out + box_out)
    for i in range(5):
        assert np.allclose(grads[i].numpy(), g['%s_gc%d' % (tag, i)], rtol=1e-5, atol=1e-8)
        assert np.allclose(grads[5 + i].numpy(), g['%s_gb%d' % (tag, i)], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_evaluation_oracle_matches_reference(golden, tag):
    """oracle/evaluation.py vs the reference's ObjectDetectionEvaluator (mAP, CorLoc, per class)"""
    from _seeded import eval_case
    from oracle import evaluation as oe
    g = golden('evaluation')
    seed, n_img, C, n_det = [int(v) for v in g[tag + '_meta']]
    ims = [dict(det_boxes=im['det_boxes'], det_scores=im['det_scores'], det_classes=im['det_classes'] - 1,
                gt_boxes=im['gt_boxes'], gt_classes=im['gt_classes'] - 1) for im in eval_case(seed, n_img, C, n_det)]
    with np.errstate(all='ignore'):
        r = oe.evaluate(ims, C)
    assert abs(r['mean_ap'] - float(g[tag + '_map'])) < 1e-12 and abs(r['mean_corloc'] - float(g[tag + '_corloc'])) < 1e-12
    assert np.allclose(r['per_class_ap'], g[tag + '_ap'], rtol=0, atol=1e-12, equal_nan=True)
    assert np.allclose(r['per_class_corloc'], g[tag + '_cl'], rtol=0, atol=1e-12, equal_nan=True)


# ---- MetaHead / AnchorNet / ProjectionNet: the oracle's restatements against the reference's own classes
#      (fixture tools/make_golden.py::gen_meta_nets; efficientdet.py:569-830)
def test_meta_head_oracle_matches_reference(golden):
    from _seeded import meta_lists, meta_nets_case, seeded_tensor
    g = golden('meta_nets')
    c = meta_nets_case(g)
    dw, pw, pb, pred, bw, bb = meta_lists(c['init'], c['extra'], c['L'], c['R'])
    tol = lambda a, b: float((a - torch.from_numpy(b)).abs().max()) <= 2e-5 * max(1.0, float(np.abs(b).max()))
    o, a = om.meta_head_forward(dw, pw, pb, bw, bb, pred, c['x'])
    for i in range(c['L']):
        assert tol(o[i], g['mh_out%d' % i]) and tol(a[i], g['mh_act%d' % i])
    o2, _ = om.meta_head_forward(dw, pw, pb, bw, bb, pred, c['x'], level_offset=2)
    assert len(o2) == c['L'] - 2 and all(tol(o2[i], g['mh_off2_out%d' % i]) for i in range(c['L'] - 2))
    # fast weights: the reference's flat list order
    fw = dw + pw + pb + pred + bw + bb
    assert len(fw) == int(g['mh_n_fast'])
    fw = [w + 0.05 * seeded_tensor(c['seed'], 'fw%d' % i, w.shape) for i, w in enumerate(fw)]
    R, L = c['R'], c['L']
    o3, a3 = om.meta_head_forward(fw[:R], fw[R:2 * R], fw[2 * R:3 * R], fw[3 * R + 3:3 * R + 3 + R * L], fw[3 * R + 3 + R * L:],
                                  fw[3 * R:3 * R + 3], c['x'])
    for i in range(L):
        assert tol(o3[i], g['mh_fw_out%d' % i]) and tol(a3[i], g['mh_fw_act%d' % i])
    # separate class head, heads='both' with the scripts' default supp_level_offset = 2 (infer.py:94)
    off = int(g['supp_level_offset_default'])
    assert int(g['mh_both_levels']) == L - off
    o4, a4, c4 = om.meta_head_forward(dw, pw, pb, bw, bb, pred, c['x'], level_offset=off,
                                      predict_class=[c['extra']['predict_pw_sep'], c['extra']['predict_pb_sep']])
    for i in range(L - off):
        assert tol(c4[i], g['mh_both_cls%d' % i]) and tol(o4[i], g['mh_both_anch%d' % i]) and tol(a4[i], g['mh_both_act%d' % i])


@pytest.mark.parametrize('layers', [3, 1])
def test_anchor_net_oracle_matches_reference(golden, layers):
    from _seeded import meta_nets_case, seeded_tensor
    g = golden('meta_nets')
    c = meta_nets_case(g)
    keys = [str(k) for k in g['an%d_keys' % layers]]
    shapes = [json.loads(str(s)) for s in g['an%d_shapes' % layers]]
    sd = {k: seeded_tensor(c['seed'] + layers, k, s) for k, s in zip(keys, shapes)}
    out = om.anchor_net_forward(sd, c['x_anchor'], c['L'], eps=float(g['an%d_eps' % layers]) or 1e-3)
    for i in range(c['L']):
        r = g['an%d_out%d' % (layers, i)]
        assert float((out[i] - torch.from_numpy(r)).abs().max()) <= 2e-5 * max(1.0, float(np.abs(r).max()))


def test_projection_net_oracle_matches_reference(golden):
    from _seeded import meta_nets_case, seeded_tensor
    g = golden('meta_nets')
    c = meta_nets_case(g)
    for depth in (2, 3, 4):
        dims = [int(v) for v in g['pn%d_dims' % depth]]
        d_in, outs = dims[-1], dims[:-1]
        ws, k = [], d_in
        for i, n in enumerate(outs):
            ws.append(seeded_tensor(c['seed'] + depth, 'proj%d' % i, (n, k)) * (1.0 / k) ** 0.5)
            k = n
        y = om.projection_forward(ws, c['x_proj'])
        r = g['pn%d_out' % depth]
        assert float((y - torch.from_numpy(r)).abs().max()) <= 2e-5 * max(1.0, float(np.abs(r).max()))
    for n in (1, 7, 200, 1024):
        med, cs = om.weighted_median(torch.from_numpy(g['wm%d_e' % n]), torch.from_numpy(g['wm%d_c' % n]))
        assert np.array_equal(med.numpy(), g['wm%d_med' % n]) and abs(float(cs) - float(g['wm%d_sum' % n])) <= 1e-5 * max(1.0, float(cs))
