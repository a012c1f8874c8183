"""Model-level parity on the GPU: EfficientDet modes, DetBenchPredict and OOD scores vs the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from _models import seeded_model
from _seeded import seeded_array
from oracle import model as om
from oracle import postprocess as op

DEV = 'cuda:0'


def _linf(a, b):
    return float((a.float().cpu() - b.float().cpu()).abs().max())


@pytest.fixture(scope='module')
def d0():
    model, cfg, nodes, sd = seeded_model('tf_efficientdet_d0', 256, 90, seed=3)
    x = torch.from_numpy(seeded_array(3, 'input', (2, 3, 256, 256)))
    with torch.no_grad():
        feats, activs = om.efficientdet_forward(sd, cfg, x, nodes, mode='fpn')
        cls_o, box_o = om.efficientdet_forward(sd, cfg, x, nodes)
    return dict(model=model, cfg=cfg, nodes=nodes, sd=sd, x=x, feats=feats, activs=activs, cls=cls_o, box=box_o)


def test_fp32_all_stages(d0):
    m = d0['model'].to(DEV).float()
    x = d0['x'].to(DEV)
    feats = m(x, mode='bb')
    for f, r in zip(feats, d0['feats']):
        assert f.shape == r.shape
        assert _linf(f, r) <= 1e-4 * max(1.0, float(r.abs().max()))
    feats2, activs = m(x, mode='fpn')
    for a, r in zip(activs, d0['activs']):
        assert a.shape == r.shape
        assert _linf(a, r) <= 1e-4 * max(1.0, float(r.abs().max()))
    cls_o, box_o = m(x)
    for a, r in zip(cls_o, d0['cls']):
        assert a.shape == r.shape and _linf(a, r) <= 1e-3
    for a, r in zip(box_o, d0['box']):
        assert a.shape == r.shape and _linf(a, r) <= 1e-3
    e_ref, m_ref = om.ood_scores(d0['cls'], 90)
    assert _linf(m.ood_energy, e_ref) <= 1e-3 and _linf(m.ood_max_logit, m_ref) <= 1e-3


def test_fp32_modes_consistent(d0):
    m = d0['model'].to(DEV).float()
    x = d0['x'].to(DEV)
    cls_full, box_full = m(x)
    cls_full = [c.clone() for c in cls_full]
    box_full = [b.clone() for b in box_full]
    feats = [f.clone() for f in m(x, mode='bb')]
    activs = [a.clone() for a in m(feats, mode='only_fpn')]
    for a, r in zip(activs, d0['activs']):
        assert _linf(a, r) <= 1e-4 * max(1.0, float(r.abs().max()))
    c2, b2 = m(feats, mode='fpn_and_head')
    assert all(torch.equal(a, b) for a, b in zip(c2, cls_full)) and all(torch.equal(a, b) for a, b in zip(b2, box_full))
    c3, b3 = m(activs, mode='head')
    assert all(torch.equal(a, b) for a, b in zip(c3, cls_full)) and all(torch.equal(a, b) for a, b in zip(b3, box_full))
    a4, b4 = m(feats, mode='not_cls')
    assert all(torch.equal(a, b) for a, b in zip(b4, box_full))
    with pytest.raises(RuntimeError):          # the MetaHead modes need model.class_net = MetaHead(...) first (infer.py:191)
        m(activs, mode='qry_cls')


def test_bf16_measured(d0):
    """bf16 throughput mode, whole network: the error of the head outputs against the fp32 oracle, against the oracle run
    on bf16-rounded weights + input (fp32 compute), and the cost of that weight rounding alone - measured, printed, bounded."""
    import copy
    m = copy.deepcopy(d0['model']).to(DEV).to(torch.bfloat16)
    x = d0['x'].to(DEV).to(torch.bfloat16)
    cls_o, box_o = m(x)
    sdq = {k: (v.to(torch.bfloat16).float() if v.is_floating_point() else v) for k, v in d0['sd'].items()}
    with torch.no_grad():
        cls_q, box_q = om.efficientdet_forward(sdq, d0['cfg'], d0['x'].to(torch.bfloat16).float(), d0['nodes'])

    def rms(a, b):
        a, b = a.float().cpu(), b.float().cpu()
        return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())
    vs_fp32 = max(rms(a, r) for a, r in zip(list(cls_o) + list(box_o), list(d0['cls']) + list(d0['box'])))
    vs_rounded = max(rms(a, r) for a, r in zip(list(cls_o) + list(box_o), list(cls_q) + list(box_q)))
    w_only = max(rms(a, r) for a, r in zip(list(cls_q) + list(box_q), list(d0['cls']) + list(d0['box'])))
    print('bf16 head outputs, rel-rms: HIP vs fp32 oracle %.3f | HIP vs oracle(bf16-rounded weights) %.3f | '
          'weight rounding alone %.3f' % (vs_fp32, vs_rounded, w_only))
    # bounds = ~1.6 x the values measured on MI355X in round 2 (0.035 / 0.077 / 0.063 on this BN-calibrated network); the
    # per-stage bands live in test_configs_gpu.py::test_bf16_per_stage_error_bounds, detection-level agreement in
    # test_bf16_detection_agreement_*
    assert vs_rounded < 0.06 and vs_fp32 < 0.12


@pytest.mark.parametrize('soft', [False, True])
def test_det_bench_predict_fp32(d0, soft):
    from ood_object_detection_amd.effdet.bench import DetBenchPredict
    m = d0['model'].to(DEV).float()
    m.config.soft_nms = soft
    bench = DetBenchPredict(m).to(DEV)
    x = d0['x'].to(DEV)
    img_info = {'img_scale': torch.tensor([1.5, 0.75], device=DEV), 'img_size': torch.tensor([[384., 384.], [190., 192.]], device=DEV)}
    for info in (None, img_info):
        det = bench(x, info)
        assert det.shape == (2, 100, 6)
        anchors = op.anchor_boxes(3, 7, 3, m.config.aspect_ratios, 4.0, (256, 256))
        # Discrete decisions (top-k membership, the 0.01 threshold, NMS) flip on 1-ulp logit changes, so the
        # post-processing chain is checked against the oracle fed the SAME head outputs (the head outputs
        # themselves are checked against the oracle in test_fp32_all_stages).
        eng = m._engine
        cls_g = [t.float().cpu() for t in eng.head_views(eng.cls_all, 90)]
        box_g = [t.float().cpu() for t in eng.head_views(eng.box_all, 4)]
        c, b, idx, cl = op.post_process(cls_g, box_g, 5, 90, 5000)
        e_ref, m_ref = om.ood_scores(cls_g, 90)
        for i in range(2):
            sc = None if info is None else info['img_scale'][i].cpu()
            sz = torch.tensor(256) if info is None else info['img_size'][i].cpu()
            ref, src = op.generate_detections(c[i], b[i], anchors, idx[i], cl[i], sc, sz, 100, soft, return_aux=True)
            n = int(bench.last_count[i])
            assert n == ref.shape[0]
            got = det[i, :n].cpu()
            assert torch.equal(got[:, 5], ref[:, 5])                                   # classes exact
            assert float((got[:, 4] - ref[:, 4]).abs().max()) <= 1e-5                  # scores (sigmoid / exp ulps)
            assert float((got[:, :4] - ref[:, :4]).abs().max()) <= 1e-3               # boxes: north-star 1e-3 px
            a_idx = idx[i][src]
            assert float((bench.last_ood['energy'][i, :n].cpu() - e_ref[i][a_idx]).abs().max()) <= 1e-4
            assert float((bench.last_ood['max_logit'][i, :n].cpu() - m_ref[i][a_idx]).abs().max()) <= 1e-5
        # end-to-end vs the all-CPU oracle: same detections up to the rare discrete flip
        c0, b0, idx0, cl0 = op.post_process(d0['cls'], d0['box'], 5, 90, 5000)
        for i in range(2):
            sc = None if info is None else info['img_scale'][i].cpu()
            sz = torch.tensor(256) if info is None else info['img_size'][i].cpu()
            ref = op.generate_detections(c0[i], b0[i], anchors, idx0[i], cl0[i], sc, sz, 100, soft)
            got = det[i, :int(bench.last_count[i])].cpu()
            matched = 0
            for r in ref:
                same = got[got[:, 5] == r[5]]
                if same.numel() and float((same[:, :4] - r[:4]).abs().max(dim=1)[0].min()) <= 0.05 + 1e-4 * float(r[:4].abs().max()):
                    matched += 1
            assert matched >= 0.9 * ref.shape[0], (matched, ref.shape[0])
    m.config.soft_nms = False


def test_d1_channels_88():
    """fpn_channels = 88 (not a multiple of the 64-byte K-chunk in bf16) and 4 BiFPN cells"""
    model, cfg, nodes, sd = seeded_model('tf_efficientdet_d1', 128, 5, seed=4)
    x = torch.from_numpy(seeded_array(4, 'input', (1, 3, 128, 128)))
    with torch.no_grad():
        cls_r, box_r = om.efficientdet_forward(sd, cfg, x, nodes)
    m = model.to(DEV).float()
    cls_o, box_o = m(x.to(DEV))
    for a, r in zip(list(cls_o) + list(box_o), list(cls_r) + list(box_r)):
        assert _linf(a, r) <= 1e-4 * max(1.0, float(r.abs().max()))


@pytest.mark.parametrize('name,size,ncls', [('tf_efficientdet_d2', 256, 7), ('tf_efficientdet_d4', 256, 4),
                                            ('tf_efficientdet_d3', 256, 5), ('tf_efficientdet_d5', 256, 90)])
def test_d2_d4_small(name, size, ncls):
    """BASELINE configs 3 / 4 use d2 and d4: fpn 112 / 224 channels, 5 / 7 BiFPN cells, 4 head repeats, deeper backbones;
    d3 (fpn 160, b3 backbone) sits between them; d5 (fpn 288, b5 backbone, 90 classes: the float32 class head takes two 64-row
    chunks per anchor because one 96-row chunk no longer fits in LDS beside the 288-channel tile)"""
    model, cfg, nodes, sd = seeded_model(name, size, ncls, seed=6)
    x = torch.from_numpy(seeded_array(6, 'input', (1, 3, size, size)))
    with torch.no_grad():
        cls_r, box_r = om.efficientdet_forward(sd, cfg, x, nodes)
    m = model.to(DEV).float()
    cls_o, box_o = m(x.to(DEV))
    for a, r in zip(list(cls_o) + list(box_o), list(cls_r) + list(box_r)):
        assert a.shape == r.shape
        assert _linf(a, r) <= 2e-4 * max(1.0, float(r.abs().max()))
    mb = model.to(torch.bfloat16)
    cls_b, box_b = mb(x.to(DEV).to(torch.bfloat16))
    assert all(bool(torch.isfinite(t.float()).all()) for t in list(cls_b) + list(box_b))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_uint8_input_equals_normalised_input(dtype):
    """A raw uint8 batch (normalisation fused into the stem kernel) gives bit-identical outputs to feeding the
    loader-normalised tensor cast to the model dtype."""
    import copy
    from ood_object_detection_amd.effdet.preprocess import normalize_batch
    model, cfg, nodes, sd = seeded_model('tf_efficientdet_d0', 128, 20, seed=2)
    model = copy.deepcopy(model).to(DEV).to(dtype)
    g = torch.Generator().manual_seed(9)
    xu = torch.randint(0, 256, (2, 3, 128, 128), generator=g, dtype=torch.uint8).to(DEV)
    with torch.no_grad():
        ca, ba = model(xu)
        ca = [t.clone() for t in ca]; ba = [t.clone() for t in ba]
        cb, bb = model(normalize_batch(xu, dtype=torch.float32).to(dtype))
    for a, b in zip(ca + ba, list(cb) + list(bb)):
        assert torch.equal(a, b)


def test_detbench_split_streams_identical():
    """Two concurrent half-batches (the default for B >= 16) give exactly the single-stream results."""
    import copy
    from ood_object_detection_amd.effdet.bench import DetBenchPredict
    model, cfg, nodes, sd = seeded_model('tf_efficientdet_d0', 128, 20, seed=2)
    model = copy.deepcopy(model).to(DEV)
    x = torch.from_numpy(seeded_array(12, 'input', (16, 3, 128, 128))).to(DEV)
    with torch.no_grad():
        b1 = DetBenchPredict(model, streams=1).to(DEV)
        d1 = b1(x).clone()
        c1, o1 = b1.last_count.clone(), {k: v.clone() for k, v in b1.last_ood.items()}
        b2 = DetBenchPredict(model).to(DEV)            # automatic: 2 streams at B = 16
        d2 = b2(x)
        torch.cuda.synchronize()
    assert torch.equal(d1, d2) and torch.equal(c1, b2.last_count)
    for k in o1:
        assert torch.equal(o1[k], b2.last_ood[k]), k
    # weights changed in place -> the shallow copies must rebuild their packed weights
    with torch.no_grad():
        model.class_net.predict.conv_pw.bias.add_(0.5)
        model.invalidate()
        d3 = b2(x).clone()
        d4 = DetBenchPredict(model, streams=1).to(DEV)(x)
    assert torch.equal(d3, d4) and not torch.equal(d3, d2)


def test_detbench_bench_size_properties():
    """The BASELINE workload shape (d0, 640 px, bf16, hard NMS) through size-independent properties: valid rows are
    sorted by score, classes are 1-based and in range, kept same-class boxes respect the NMS threshold, and the OOD
    scores obey -energy >= max_logit (logsumexp >= max)."""
    import bench as B
    from ood_object_detection_amd.effdet.bench import DetBenchPredict
    model = B.build_model('tf_efficientdet_d0', 640, 90).to(DEV).to(torch.bfloat16)
    bench = DetBenchPredict(model).to(DEV)
    x = torch.randn(16, 3, 640, 640, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1)).to(torch.bfloat16)
    with torch.no_grad():
        det = bench(x).float().cpu()
    cnt = bench.last_count.cpu()
    assert det.shape == (16, 100, 6) and int(cnt.max()) <= 100 and int(cnt.min()) > 0
    for i in range(16):
        d = det[i, :int(cnt[i])]
        assert torch.all(d[:-1, 4] >= d[1:, 4]) and torch.all(d[:, 4] > 0.01)
        assert torch.all((d[:, 5] >= 1) & (d[:, 5] <= 90)) and torch.all(d[:, 2] >= d[:, 0]) and torch.all(d[:, 3] >= d[:, 1])
        assert torch.all(det[i, int(cnt[i]):] == 0)
        x1 = torch.max(d[:, None, 0], d[None, :, 0]); y1 = torch.max(d[:, None, 1], d[None, :, 1])
        x2 = torch.min(d[:, None, 2], d[None, :, 2]); y2 = torch.min(d[:, None, 3], d[None, :, 3])
        inter = (x2 - x1).clamp(min=0) * (y2 - y1).clamp(min=0)
        area = (d[:, 2] - d[:, 0]) * (d[:, 3] - d[:, 1])
        iou = inter / (area[:, None] + area[None, :] - inter).clamp(min=1e-9)
        same = (d[:, None, 5] == d[None, :, 5]) & ~torch.eye(d.shape[0], dtype=torch.bool)
        assert (float(iou[same].max()) <= 0.3 + 1e-3) if same.any() else True      # hard NMS threshold (anchors.py:150)
    ood = bench.last_ood
    assert torch.isfinite(ood['anchor_energy']).all() and torch.all(-ood['anchor_energy'] >= ood['anchor_max_logit'] - 1e-4)
    assert torch.isfinite(ood['energy'][:, :1]).all()


def _meta_head(dtype, seed=5):
    from ood_object_detection_amd.effdet.meta_head import MetaHead
    model, cfg, nodes, sd = seeded_model('tf_efficientdet_d0', 128, 20, seed=2)
    init = {k: v for k, v in model.state_dict().items() if k.startswith('class_net.')}
    torch.manual_seed(seed)
    mh = MetaHead(model.config, pretrain_init=init)
    with torch.no_grad():                       # non-trivial BN affine parameters
        for n, p in mh.named_parameters():
            if n.startswith('bn_w'):
                p.copy_(1.0 + 0.2 * torch.randn_like(p))
            if n.startswith('bn_b'):
                p.copy_(0.1 * torch.randn_like(p))
    return model, mh.to(DEV).to(dtype)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_meta_head_forward(dtype):
    """MetaHead (functional class head with batch-statistics BN) vs the oracle restatement of efficientdet.py:636-695:
    outputs, x_pred activations, fast_weights path, level_offset and the separate class head."""
    model, mh = _meta_head(dtype)
    F_, B = model.config.fpn_channels, 3
    sizes = [16, 8, 4, 2, 1]
    x = [torch.from_numpy(seeded_array(31, 'lvl%d' % i, (B, F_, s, s))).to(DEV).to(dtype) for i, s in enumerate(sizes)]
    xr = [t.float().cpu() for t in x]

    def close(a, b):
        a = a.float().cpu()
        if dtype == torch.float32:
            return float((a - b).abs().max()) <= 2e-4 * max(1.0, float(b.abs().max()))
        # bf16 (measured, not assumed): batch statistics over a handful of pixels (the 1x1 and 2x2 levels) amplify the
        # rounding noise, so the check is on the relative rms error of the whole tensor
        return float((a - b).norm()) <= 0.05 * max(float(b.norm()), 1e-3)

    P = lambda ps: [p.detach().float().cpu() for p in ps]
    with torch.no_grad():
        outs, activs = mh(x, ret_activs=True)
        ro, ra = om.meta_head_forward(P(mh.conv_dw_rep), P(mh.conv_pw_rep), P(mh.conv_pb_rep), P(mh.bn_rep_w), P(mh.bn_rep_b), P(mh.predict), xr)
        for a, b in zip(outs + activs, ro + ra):
            assert tuple(a.shape) == tuple(b.shape)
            assert close(a, b)
        # fast weights (reference list order) = the module's own parameters -> same result; level_offset drops levels
        fw = mh.conv_dw_rep + mh.conv_pw_rep + mh.conv_pb_rep + mh.predict + mh.bn_rep_w + mh.bn_rep_b
        outs2 = mh(x, fast_weights=fw)
        for a, b in zip(outs, outs2):
            assert torch.equal(a, b)
        outs3 = mh(x, level_offset=2)
        assert len(outs3) == 3 and all(torch.equal(a, b) for a, b in zip(outs3, outs[2:]))
        # separate class head on the same x_pred (heads='both')
        mh.add_head()
        mh.to(DEV).to(dtype)
        co, ao, act = mh(x, ret_activs=True, heads='both')
        rco = om.meta_head_forward(P(mh.conv_dw_rep), P(mh.conv_pw_rep), P(mh.conv_pb_rep), P(mh.bn_rep_w), P(mh.bn_rep_b), P(mh.predict), xr,
                                   predict_class=P(mh.predict_class))[2]
        for a, b in zip(co, rco):
            assert close(a, b)
        assert all(torch.equal(a, b) for a, b in zip(ao, outs))


def test_meta_head_modes_through_model():
    """EfficientDet.forward(mode='qry_cls' / 'supp_cls') with model.class_net = MetaHead(...) (infer.py:191,359,681) on the
    model's own BiFPN outputs (zero-copy pyramid views)."""
    import copy
    model, mh = _meta_head(torch.float32)
    model = copy.deepcopy(model).to(DEV)
    x = torch.from_numpy(seeded_array(32, 'img', (2, 3, 128, 128))).to(DEV)
    with torch.no_grad():
        activs = model(x, mode='supp_bb')
        model.class_net = mh
        q = model(activs, mode='qry_cls')
        P = lambda ps: [p.detach().float().cpu() for p in ps]
        ro, ra = om.meta_head_forward(P(mh.conv_dw_rep), P(mh.conv_pw_rep), P(mh.conv_pb_rep), P(mh.bn_rep_w), P(mh.bn_rep_b), P(mh.predict),
                                      [a.float().cpu() for a in activs])
        for a, b in zip(q, ro):
            assert float((a.float().cpu() - b).abs().max()) <= 2e-4 * max(1.0, float(b.abs().max()))
        mh.add_head(); mh.to(DEV)
        co, ao, act = model(activs, mode='supp_cls')
        assert len(co) == len(ao) == len(act) == 3         # supp_level_offset = 2 (infer.py:94)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('layers', [3, 1])
def test_anchor_net_forward(dtype, layers):
    """AnchorNet (HeadNet-shaped tower to 9 outputs per cell) vs the oracle restatement."""
    from ood_object_detection_amd.effdet.aux_nets import AnchorNet
    model, cfg, nodes, sd0 = seeded_model('tf_efficientdet_d0', 128, 20, seed=2)
    torch.manual_seed(3)
    net = AnchorNet(model.config, num_anch_layers=layers).eval()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.uniform_(0.8, 1.2); m.bias.normal_(0, 0.1)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.to(DEV).to(dtype)
    F_, B = model.config.fpn_channels, 2
    x = [torch.from_numpy(seeded_array(41, 'lvl%d' % i, (B, F_, s, s))).to(DEV).to(dtype) for i, s in enumerate([16, 8, 4, 2, 1])]
    with torch.no_grad():
        out = net(x)
        ref = om.anchor_net_forward(sd, [t.float().cpu() for t in x], 5, eps=net.bn_rep[0][0].bn.eps if layers > 1 else 1e-3)
    for a, b in zip(out, ref):
        assert tuple(a.shape) == tuple(b.shape)
        err = float((a.float().cpu() - b).abs().max())
        assert err <= (2e-4 if dtype == torch.float32 else 8e-2) * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_projection_net(dtype):
    """ProjectionNet MLP (K = 106 zero-padded to 112 for the GEMM), encoding tables and weighted_median vs the oracle."""
    from ood_object_detection_amd.effdet.aux_nets import ProjectionNet
    model, cfg, nodes, sd0 = seeded_model('tf_efficientdet_d0', 128, 20, seed=2)
    torch.manual_seed(6)
    net = ProjectionNet(model.config, width=128, proj_depth=3)
    assert tuple(net.anch_enc.shape) == (9, 8) and tuple(net.cell_enc.shape) == (80, 14) and tuple(net.lev_enc.shape) == (5, 6)
    ws = [m.weight.detach().clone() for m in net.projection if isinstance(m, torch.nn.Linear)]
    net = net.to(DEV).to(dtype)
    x = torch.from_numpy(seeded_array(43, 'px', (5, 37, 64 + 42))).to(DEV).to(dtype)
    with torch.no_grad():
        y = net(x)
        ref = om.projection_forward([w.to(dtype).float() for w in ws], x.float().cpu())
    assert tuple(y.shape) == (5, 37, 64)
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    assert float((y.float().cpu() - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))
    if dtype == torch.float32:
        g = torch.Generator().manual_seed(2)
        for n in (1, 7, 200, 1024):
            e = torch.randn(n, 24, generator=g)
            c = torch.rand(n, generator=g)
            med, cs = net.weighted_median(e.to(DEV), c.to(DEV))
            rm, rc = om.weighted_median(e, c)
            assert torch.equal(med.cpu(), rm) and abs(float(cs) - float(rc)) <= 1e-4 * max(1.0, float(rc))


def test_baseline_config1_d0_512_uint8_images():
    """BASELINE configs[0]: tf_efficientdet_d0, 512 x 512, batch 1, float32, eight synthetic uint8
    images `randint(0, 256, (3, 512, 512), seed = i)` normalised with the ImageNet constants - DetBenchPredict on the GPU
    (raw uint8 in, normalisation fused into the first kernel) against the CPU oracle path image by image: head outputs
    <= 1e-3 abs (north star), detections through the same-logits stage check, OOD scores <= 1e-4."""
    import copy
    from oracle import preprocess as opre
    from ood_object_detection_amd.effdet.bench import DetBenchPredict
    model, cfg, nodes, sd = seeded_model('tf_efficientdet_d0', 512, 90, seed=11)
    m = copy.deepcopy(model).to(DEV).float()
    bench = DetBenchPredict(m).to(DEV)
    anchors = op.anchor_boxes(3, 7, 3, m.config.aspect_ratios, 4.0, (512, 512))
    assert anchors.shape[0] == 49104                                     # N at 512 px
    for i in range(8):
        xu = torch.randint(0, 256, (3, 512, 512), generator=torch.Generator().manual_seed(i), dtype=torch.uint8)[None]
        xn = torch.from_numpy(opre.normalize_u8(xu.numpy()))
        with torch.no_grad():
            cls_r, box_r = om.efficientdet_forward(sd, cfg, xn, nodes)
            det = bench(xu.to(DEV))
        eng = m._engine
        cls_g = [t.float().cpu() for t in eng.head_views(eng.cls_all, 90)]
        box_g = [t.float().cpu() for t in eng.head_views(eng.box_all, 4)]
        assert max(float((a - b).abs().max()) for a, b in zip(cls_g, cls_r)) <= 1e-3
        assert max(float((a - b).abs().max()) for a, b in zip(box_g, box_r)) <= 1e-3
        c, b, idx, cl = op.post_process(cls_g, box_g, 5, 90, 5000)
        ref, src = op.generate_detections(c[0], b[0], anchors, idx[0], cl[0], None, torch.tensor(512), 100, False, return_aux=True)
        n = int(bench.last_count[0])
        assert n == ref.shape[0]
        got = det[0, :n].cpu()
        if n:
            assert torch.equal(got[:, 5], ref[:, 5])
            assert float((got[:, 4] - ref[:, 4]).abs().max()) <= 1e-5 and float((got[:, :4] - ref[:, :4]).abs().max()) <= 1e-3
        e_ref, m_ref = om.ood_scores(cls_r, 90)
        assert float((m.ood_energy.cpu() - e_ref).abs().max()) <= 1e-4 * max(1.0, float(e_ref.abs().max()))
        assert float((m.ood_max_logit.cpu() - m_ref).abs().max()) <= 1e-3


@pytest.mark.parametrize('graphs', [False, True])
def test_pipelined_predict_identical_to_sequential(graphs):
    """serving.PipelinedPredict: three batches in flight over shared weights (eager launches or one hipGraph per slot) give
    bit-identical detections / OOD scores"""
    from _models import seeded_model
    from ood_object_detection_amd.effdet.bench import DetBenchPredict
    from ood_object_detection_amd.serving import PipelinedPredict
    model, cfg, nodes, sd = seeded_model('tf_efficientdet_d0', 128, 20, seed=11, cls_bias=0.0)
    model = model.to(DEV).to(torch.bfloat16)
    g = torch.Generator().manual_seed(4)
    batches = [torch.randn(4, 3, 128, 128, generator=g).to(DEV).to(torch.bfloat16) for _ in range(7)]
    ref_bench = DetBenchPredict(model, streams=1).to(DEV)
    refs = []
    with torch.no_grad():
        for x in batches:
            det = ref_bench(x)
            refs.append((det.clone(), ref_bench.last_count.clone(), ref_bench.last_ood['energy'].clone()))
    pipe = PipelinedPredict(model, in_flight=3, graphs=graphs)
    tickets, outs = [], {}
    for i, x in enumerate(batches):
        if i >= 3:
            outs[tickets[i - 3]] = pipe.result(tickets[i - 3])
        tickets.append(pipe.submit(x))
    for t in tickets[-3:]:
        outs[t] = pipe.result(t)
    with pytest.raises(KeyError):
        pipe.result(tickets[0])
    for t, (det_r, cnt_r, en_r) in zip(tickets, refs):
        det, cnt, ood = outs[t]
        assert torch.equal(cnt, cnt_r) and torch.equal(det, det_r) and torch.equal(ood['energy'], en_r)
    assert int(refs[0][1].sum()) > 0


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_one_class_head_ood_degenerates_to_the_logit(dtype):
    """with C = 1 (infer.py:192) energy = -z and max-logit = z exactly; detections still come out"""
    from ood_object_detection_amd.effdet.bench import DetBenchPredict
    model, cfg, nodes, sd = seeded_model('tf_efficientdet_d0', 256, 1, seed=13, cls_bias=0.0)
    model = model.to(DEV).to(dtype)
    x = torch.from_numpy(seeded_array(13, 'input', (2, 3, 256, 256))).to(DEV).to(dtype)
    with torch.no_grad():
        cls_o, box_o = model(x)
        z = torch.cat([c.permute(0, 2, 3, 1).reshape(2, -1) for c in cls_o], 1).float()
        assert z.shape == tuple(model.ood_max_logit.shape)
        # the scores are float32 values of the accumulator; the stored logits are rounded to the model dtype
        assert torch.equal(model.ood_max_logit.to(dtype).float(), z)
        assert torch.allclose(model.ood_energy, -model.ood_max_logit, rtol=0, atol=2e-6)
        det = DetBenchPredict(model).to(DEV)(x)
    assert det.shape == (2, 100, 6) and bool(torch.isfinite(det).all())
    assert set(det[..., 5].unique().tolist()) <= {0.0, 1.0}


def test_det_bench_predict_ignores_training_mode_and_grad_mode():
    """a bf16 model left in training mode, called without torch.no_grad(): DetBenchPredict still runs the inference engine
    (the differentiable path is float32 only and would raise) and gives the same detections as the eval / no_grad call"""
    from ood_object_detection_amd.effdet.bench import DetBenchPredict
    model, cfg, nodes, sd = seeded_model('tf_efficientdet_d0', 128, 20, seed=11, cls_bias=0.0)
    model = model.to(DEV).to(torch.bfloat16)
    x = torch.from_numpy(seeded_array(11, 'input', (2, 3, 128, 128))).to(DEV).to(torch.bfloat16)
    bench = DetBenchPredict(model, streams=1).to(DEV)
    with torch.no_grad():
        ref = bench(x).clone()
    model.train()
    assert model.wants_autograd()
    det = bench(x)
    assert not det.requires_grad and torch.equal(det, ref)
    with pytest.raises(RuntimeError):
        model(x)                      # the differentiable path itself refuses bfloat16


def test_wider_than_d5_is_refused_with_a_reason():
    """widths above 288 (d6 / d7: 384) do not fit the fused kernel's LDS tile: the engine says so instead of failing in a launch"""
    from ood_object_detection_amd.effdet.config import get_efficientdet_config
    from ood_object_detection_amd.effdet.efficientdet import EfficientDet
    h = get_efficientdet_config('tf_efficientdet_d5')
    h.fpn_channels = 384
    h.image_size = (256, 256)
    h.num_classes = 3
    m = EfficientDet(h, pretrained_backbone=False).to(DEV).eval()
    with pytest.raises(NotImplementedError):
        with torch.no_grad():
            m(torch.zeros(1, 3, 256, 256, device=DEV))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_non_square_input(dtype):
    """H != W (128 x 256): every kernel carries both extents; head outputs against the oracle, detections come out"""
    from ood_object_detection_amd.effdet.bench import DetBenchPredict
    model, cfg, nodes, sd = seeded_model('tf_efficientdet_d0', 128, 12, seed=17, cls_bias=0.0)
    x = torch.from_numpy(seeded_array(17, 'input', (2, 3, 128, 256)))
    with torch.no_grad():
        cls_r, box_r = om.efficientdet_forward(sd, cfg, x, nodes)
    m = model.to(DEV).to(dtype)
    with torch.no_grad():
        cls_o, box_o = m(x.to(DEV).to(dtype))
        for a, r in zip(list(cls_o) + list(box_o), list(cls_r) + list(box_r)):
            assert a.shape == r.shape
            if dtype == torch.float32:
                assert _linf(a, r) <= 1e-4 * max(1.0, float(r.abs().max()))
            else:
                assert bool(torch.isfinite(a.float()).all())
        det = DetBenchPredict(m).to(DEV)(x.to(DEV).to(dtype))
    assert det.shape == (2, 100, 6) and float(det[..., 2].max()) > 128.0      # boxes reach into the wide half


# ---- MetaHead / AnchorNet / ProjectionNet against the reference's own classes (fixture meta_nets.npz, generated by
#      tools/make_golden.py from effdet/efficientdet.py:569-830 run on the CPU)
def _close_ref(a, r, tol=2e-4):
    r = torch.from_numpy(np.asarray(r))
    a = a.float().cpu()
    return tuple(a.shape) == tuple(r.shape) and float((a - r).abs().max()) <= tol * max(1.0, float(r.abs().max()))


def test_meta_head_matches_reference_fixture(golden):
    from _seeded import meta_lists, meta_nets_case, seeded_tensor
    from ood_object_detection_amd.effdet.config import get_efficientdet_config
    from ood_object_detection_amd.effdet.meta_head import MetaHead
    g = golden('meta_nets')
    c = meta_nets_case(g)
    cfg = get_efficientdet_config('tf_efficientdet_d0')
    mh = MetaHead(cfg, pretrain_init=c['init'])
    with torch.no_grad():
        mh.predict_pw.copy_(c['extra']['predict_pw']); mh.predict_pb.copy_(c['extra']['predict_pb'])
    mh = mh.to(DEV)
    x = [t.to(DEV) for t in c['x']]
    L = c['L']
    with torch.no_grad():
        o, a = mh(x, ret_activs=True)
        for i in range(L):
            assert _close_ref(o[i], g['mh_out%d' % i]) and _close_ref(a[i], g['mh_act%d' % i])
        o2 = mh(x, level_offset=2)
        assert len(o2) == L - 2 and all(_close_ref(o2[i], g['mh_off2_out%d' % i]) for i in range(L - 2))
        fw = mh.conv_dw_rep + mh.conv_pw_rep + mh.conv_pb_rep + mh.predict + mh.bn_rep_w + mh.bn_rep_b
        fw = [w.detach() + 0.05 * seeded_tensor(c['seed'], 'fw%d' % i, w.shape).to(DEV) for i, w in enumerate(fw)]
        o3, a3 = mh(x, fast_weights=fw, ret_activs=True)
        for i in range(L):
            assert _close_ref(o3[i], g['mh_fw_out%d' % i]) and _close_ref(a3[i], g['mh_fw_act%d' % i])
        mh.add_head()
        mh.predict_pw_sep.data.copy_(c['extra']['predict_pw_sep']); mh.predict_pb_sep.data.copy_(c['extra']['predict_pb_sep'])
        mh.to(DEV)
        off = int(g['supp_level_offset_default'])
        co, ao, act = mh(x, ret_activs=True, level_offset=off, heads='both')
        assert len(co) == int(g['mh_both_levels']) == L - off
        for i in range(L - off):
            assert _close_ref(co[i], g['mh_both_cls%d' % i]) and _close_ref(ao[i], g['mh_both_anch%d' % i]) and _close_ref(act[i], g['mh_both_act%d' % i])


def test_supp_cls_mode_uses_the_scripts_default_level_offset(golden):
    """EfficientDet.forward(mode='supp_cls') under defaults returns num_levels - 2 outputs (FLAGS.supp_level_offset = 2,
    infer.py:94 / pretrain.py:63) that line up with the 3-level proj_anchors of dataloader.py:66"""
    import copy
    model, mh = _meta_head(torch.float32)
    model = copy.deepcopy(model).to(DEV)
    x = torch.from_numpy(seeded_array(32, 'img', (2, 3, 128, 128))).to(DEV)
    with torch.no_grad():
        activs = model(x, mode='supp_bb')
        mh.add_head(); mh.to(DEV)
        model.class_net = mh
        co, ao, act = model(activs, mode='supp_cls')
        assert len(co) == len(ao) == len(act) == model.config.num_levels - 2 == 3
        assert [t.shape[-1] for t in ao] == [a.shape[-1] for a in activs[2:]]
        model.supp_level_offset = 0
        assert len(model(activs, mode='supp_cls')[0]) == 5


@pytest.mark.parametrize('layers', [3, 1])
def test_anchor_net_matches_reference_fixture(golden, layers):
    import json
    from _seeded import meta_nets_case, seeded_tensor
    from ood_object_detection_amd.effdet.aux_nets import AnchorNet
    from ood_object_detection_amd.effdet.config import get_efficientdet_config
    g = golden('meta_nets')
    c = meta_nets_case(g)
    cfg = get_efficientdet_config('tf_efficientdet_d0')
    net = AnchorNet(cfg, num_anch_layers=layers).eval()
    sd = net.state_dict()
    new = {k: (v if k.endswith('num_batches_tracked') else seeded_tensor(c['seed'] + layers, k, v.shape)) for k, v in sd.items()}
    net.load_state_dict(new, strict=True)
    net = net.to(DEV)
    with torch.no_grad():
        out = net([t.to(DEV) for t in c['x_anchor']])
    for i in range(c['L']):
        assert _close_ref(out[i], g['an%d_out%d' % (layers, i)])


def test_projection_net_matches_reference_fixture(golden):
    from _seeded import meta_nets_case, seeded_tensor
    from ood_object_detection_amd.effdet.aux_nets import ProjectionNet
    from ood_object_detection_amd.effdet.config import get_efficientdet_config
    g = golden('meta_nets')
    c = meta_nets_case(g)
    cfg = get_efficientdet_config('tf_efficientdet_d0')
    for depth in (2, 3, 4):
        net = ProjectionNet(cfg, 128, proj_depth=depth)
        with torch.no_grad():
            for i, m in enumerate([m for m in net.projection if isinstance(m, torch.nn.Linear)]):
                m.weight.copy_(seeded_tensor(c['seed'] + depth, 'proj%d' % i, m.weight.shape) * (1.0 / m.in_features) ** 0.5)
            y = net.to(DEV)(c['x_proj'].to(DEV))
        assert _close_ref(y, g['pn%d_out' % depth], tol=2e-5)
    for n in (1, 7, 200, 1024):
        med, cs = net.weighted_median(torch.from_numpy(g['wm%d_e' % n]).to(DEV), torch.from_numpy(g['wm%d_c' % n]).to(DEV))
        assert np.array_equal(med.cpu().numpy(), g['wm%d_med' % n])
        assert abs(float(cs) - float(g['wm%d_sum' % n])) <= 1e-4 * max(1.0, float(g['wm%d_sum' % n]))


def test_engine_follows_in_place_parameter_updates():
    """after a stock optimizer step (no `invalidate()` call) an inference forward must run on the NEW weights: the
    engine key carries a fingerprint of the parameters' version counters, and train() <-> eval() transitions drop the engine"""
    import copy
    model, cfg, nodes, sd = seeded_model('tf_efficientdet_d0', 128, 20, seed=11)
    model = copy.deepcopy(model).to(DEV)
    x = torch.from_numpy(seeded_array(11, 'input', (2, 3, 128, 128))).to(DEV)

    def fwd(m):
        with torch.no_grad():
            c, b = m(x)
            return [t.clone() for t in list(c) + list(b)]

    def fresh():
        f = copy.deepcopy(model)
        f.invalidate()
        return fwd(f)
    y0 = fwd(model)
    eng0 = model._engine
    assert fwd(model)[0].data_ptr() != 0 and model._engine is eng0                  # unchanged weights: the engine is reused
    with torch.no_grad():
        for p in model.class_net.parameters():
            p.add_(0.01)                                                             # in place, no invalidate()
    y1 = fwd(model)
    assert model._engine is not eng0
    assert all(torch.equal(a, b) for a, b in zip(y1, fresh())) and not torch.equal(y0[0], y1[0])
    # stock torch.optim.Adam + clip_grad_norm_, then model.eval() forward
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    for p in model.parameters():
        p.grad = torch.ones_like(p) * 0.1
    torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
    opt.step()
    y2 = fwd(model)
    assert all(torch.equal(a, b) for a, b in zip(y2, fresh())) and not torch.equal(y1[0], y2[0])
    # BatchNorm running statistics written in place
    with torch.no_grad():
        model.fpn.cell[0].fnode[0].after_combine.conv.bn.running_var.mul_(1.5)
    y3 = fwd(model)
    assert all(torch.equal(a, b) for a, b in zip(y3, fresh())) and not torch.equal(y2[0], y3[0])
    # train() / eval() transitions rebuild as well
    e = model._engine
    model.train(); model.eval()
    fwd(model)
    assert model._engine is not e


def test_nested_fork_inside_capture_is_refused():
    """The round-1 segfault (box head forked onto a third stream inside a captured half-batch -> hipStreamEndCapture crash)
    is unreachable through the package: DetBenchPredict refuses to fork from a forked stream while capturing.  Forking from
    the capture's origin stream (what bench.py --sub-batches 0 and PipelinedPredict do) still works."""
    from ood_object_detection_amd.effdet.bench import DetBenchPredict, forked_stream
    model, cfg, nodes, sd = seeded_model('tf_efficientdet_d0', 128, 20, seed=11, cls_bias=0.0)
    model = model.to(DEV).to(torch.bfloat16)
    x = torch.from_numpy(seeded_array(11, 'input', (4, 3, 128, 128))).to(DEV).to(torch.bfloat16)
    bench = DetBenchPredict(model, streams=2).to(DEV)
    with torch.no_grad():
        ref = bench(x).clone()
        side, inner = torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            out = bench(x)                                  # fork from the origin stream: fine
            cur = torch.cuda.current_stream(DEV)
            inner.wait_stream(cur)
            with forked_stream(inner):
                with pytest.raises(RuntimeError, match='nested forks'):
                    bench(x)                                # fork from a fork while capturing: refused before any launch
            cur.wait_stream(inner)
        g.replay()
        torch.cuda.synchronize()
    assert torch.equal(out, ref)


@pytest.mark.parametrize('both,offset,first_order', [(False, 0, False), (True, 2, False), (False, 0, True), (True, 2, True)])
def test_meta_head_gradients_match_oracle_autograd(golden, both, offset, first_order):
    """the MAML inner loop's gradients (infer.py:658: autograd.grad of a loss on the MetaHead outputs w.r.t. the head's
    parameters; :681 the query pass with fast weights): d loss / d (every weight, every input level) from the HIP training
    kernels against torch autograd through the oracle's MetaHead restatement (itself pinned to the reference class)"""
    from _seeded import meta_lists, meta_nets_case, seeded_tensor
    from ood_object_detection_amd.effdet.config import get_efficientdet_config
    from ood_object_detection_amd.effdet.meta_head import MetaHead
    g = golden('meta_nets')
    c = meta_nets_case(g)
    cfg = get_efficientdet_config('tf_efficientdet_d0')
    L, R = c['L'], c['R']
    mh = MetaHead(cfg, pretrain_init=c['init'])
    with torch.no_grad():
        mh.predict_pw.copy_(c['extra']['predict_pw']); mh.predict_pb.copy_(c['extra']['predict_pb'])
    if both:
        mh.add_head()
        mh.predict_pw_sep.data.copy_(c['extra']['predict_pw_sep']); mh.predict_pb_sep.data.copy_(c['extra']['predict_pb_sep'])
    mh = mh.to(DEV)
    mh.first_order = first_order             # True: the single-node backward (effdet/meta_grad.py); default: primitives (meta_ops.py)
    xs = [t.clone().to(DEV).requires_grad_() for t in c['x']]
    nlev = L - offset
    # a loss that reaches every output: random cotangents for outputs, x_pred activations (and the separate class head)
    go = [seeded_tensor(61, 'go%d' % i, (c['B'], 9, s, s)) for i, s in enumerate(c['sizes'][offset:])]
    ga = [seeded_tensor(61, 'ga%d' % i, (c['B'], c['F'], s, s)) * 0.3 for i, s in enumerate(c['sizes'][offset:])]
    gc = [seeded_tensor(61, 'gc%d' % i, (c['B'], 9, s, s)) for i, s in enumerate(c['sizes'][offset:])]

    def loss_of(outs, acts, couts):
        t = sum((o * w.to(o.device)).sum() for o, w in zip(outs, go)) + sum((a * w.to(a.device)).sum() for a, w in zip(acts, ga))
        if couts is not None:
            t = t + sum((o * w.to(o.device)).sum() for o, w in zip(couts, gc))
        return t
    # ---- HIP: module parameters, then again through explicit fast weights
    params = list(mh.parameters())
    if both:
        co, ao, act = mh(xs, ret_activs=True, level_offset=offset, heads='both')
    else:
        (ao, act), co = mh(xs, ret_activs=True, level_offset=offset), None
    assert len(ao) == nlev and ao[0].requires_grad
    grads = torch.autograd.grad(loss_of(ao, act, co), params + xs[offset:], allow_unused=True)
    names = [n for n, _ in mh.named_parameters()]
    # ---- oracle on the CPU
    dw, pw, pb, pred, bw, bb = meta_lists(c['init'], c['extra'], L, R)
    leaf = lambda ts: [t.clone().requires_grad_() for t in ts]
    dw, pw, pb, pred, bw, bb = leaf(dw), leaf(pw), leaf(pb), leaf(pred), leaf(bw), leaf(bb)
    pc = leaf([c['extra']['predict_pw_sep'], c['extra']['predict_pb_sep']]) if both else None
    xr = [t.clone().requires_grad_() for t in c['x']]
    res = om.meta_head_forward(dw, pw, pb, bw, bb, pred, xr, level_offset=offset, predict_class=pc)
    ro, ra, rc = (res[0], res[1], res[2]) if both else (res[0], res[1], None)
    ref_named = {}
    for r in range(R):
        ref_named['conv_dw%d' % r], ref_named['conv_pw%d' % r], ref_named['conv_pb%d' % r] = dw[r], pw[r], pb[r]
    ref_named['predict_dw'], ref_named['predict_pw'], ref_named['predict_pb'] = pred
    for lev in range(L):
        for r in range(R):
            ref_named['bn_w%d%d' % (r, lev)], ref_named['bn_b%d%d' % (r, lev)] = bw[lev * R + r], bb[lev * R + r]
    if both:
        ref_named['predict_pw_sep'], ref_named['predict_pb_sep'] = pc
    ref_grads = torch.autograd.grad(loss_of(ro, ra, rc), [ref_named[n] for n in names] + xr[offset:], allow_unused=True)
    gmax = max(float(r.abs().max()) for r in ref_grads if r is not None)
    worst = []
    for n, a, r in zip(names + ['x%d' % i for i in range(offset, L)], grads, ref_grads):
        if r is None:                                   # levels below the offset do not reach the loss
            assert a is None or float(a.abs().max()) == 0.0, n
            continue
        assert a is not None, n
        floor = 1e-5 * gmax
        if n.startswith('conv_pb'):
            floor = 1e-4 * gmax                         # bias in front of a batch-statistics BN: analytically zero gradient
        worst.append((float((a.cpu() - r).abs().max()) / max(float(r.abs().max()), floor), n))
    worst.sort(reverse=True)
    assert worst[0][0] <= 2e-3, worst[:6]
    # the fast-weight path (reference list order) gives the same gradients as the parameter path
    fw = [p.detach().clone().requires_grad_() for p in (mh.conv_dw_rep + mh.conv_pw_rep + mh.conv_pb_rep + mh.predict + mh.bn_rep_w + mh.bn_rep_b)]
    out_f = mh([t.detach() for t in xs], fast_weights=fw)
    gf = torch.autograd.grad(sum((o * w.to(DEV)).sum() for o, w in zip(out_f, [seeded_tensor(61, 'go%d' % i, (c['B'], 9, s, s)) for i, s in enumerate(c['sizes'])])), fw, allow_unused=True)
    assert all(x is not None for x in gf[:3 * R + 3])
    # one SGD inner step on the fast weights changes the query output (infer.py:660-681)
    with torch.no_grad():
        fw2 = [w - 0.1 * (gg if gg is not None else torch.zeros_like(w)) for w, gg in zip(fw, gf)]
        out_q = mh([t.detach() for t in xs], fast_weights=fw2)
        assert not torch.equal(out_q[0], out_f[0].detach())


@pytest.mark.parametrize('offset', [0, 2])
def test_meta_head_second_order_matches_oracle_autograd(golden, offset):
    """MAML as the reference runs it (infer.py:658-687): the inner gradient is taken with create_graph=True and the outer loss
    is differentiated THROUGH it.  Here: g = d L1 / d params (create_graph), then d <g, v> / d (params, inputs) - a
    Hessian-vector product with random v - from the HIP primitives (effdet/meta_ops.py) against torch autograd through the
    oracle's MetaHead on the CPU; and the full inner-step / query-loss composition of infer.py:660-687."""
    from _seeded import meta_lists, meta_nets_case, seeded_tensor
    from ood_object_detection_amd.effdet.config import get_efficientdet_config
    from ood_object_detection_amd.effdet.meta_head import MetaHead
    g = golden('meta_nets')
    c = meta_nets_case(g)
    cfg = get_efficientdet_config('tf_efficientdet_d0')
    L, R = c['L'], c['R']
    mh = MetaHead(cfg, pretrain_init=c['init'])
    with torch.no_grad():
        mh.predict_pw.copy_(c['extra']['predict_pw']); mh.predict_pb.copy_(c['extra']['predict_pb'])
    mh = mh.to(DEV)
    xs = [t.clone().to(DEV).requires_grad_() for t in c['x']]
    go = [seeded_tensor(71, 'go%d' % i, (c['B'], 9, s, s)) for i, s in enumerate(c['sizes'][offset:])]

    def inner_loss(outs):                      # non-linear in the outputs, like the BCE of infer.py:656
        return sum(torch.nn.functional.binary_cross_entropy_with_logits(o, torch.sigmoid(w.to(o.device))) for o, w in zip(outs, go))
    names = [n for n, _ in mh.named_parameters()]
    params = list(mh.parameters())
    vs = [seeded_tensor(72, 'v_' + n, tuple(p.shape)) for n, p in zip(names, params)]
    # ---- HIP
    outs = mh(xs, level_offset=offset)
    g1 = torch.autograd.grad(inner_loss(outs), params, create_graph=True, allow_unused=True)
    assert all(gi is None or gi.requires_grad for gi in g1)
    dot = sum((gi * v.to(DEV)).sum() for gi, v in zip(g1, vs) if gi is not None)
    hv = torch.autograd.grad(dot, params + xs[offset:], allow_unused=True)
    # ---- oracle on the CPU
    dw, pw, pb, pred, bw, bb = meta_lists(c['init'], c['extra'], L, R)
    leaf = lambda ts: [t.clone().requires_grad_() for t in ts]
    dw, pw, pb, pred, bw, bb = leaf(dw), leaf(pw), leaf(pb), leaf(pred), leaf(bw), leaf(bb)
    xr = [t.clone().requires_grad_() for t in c['x']]
    ro = om.meta_head_forward(dw, pw, pb, bw, bb, pred, xr, level_offset=offset)[0]
    ref_named = {}
    for r in range(R):
        ref_named['conv_dw%d' % r], ref_named['conv_pw%d' % r], ref_named['conv_pb%d' % r] = dw[r], pw[r], pb[r]
    ref_named['predict_dw'], ref_named['predict_pw'], ref_named['predict_pb'] = pred
    for lev in range(L):
        for r in range(R):
            ref_named['bn_w%d%d' % (r, lev)], ref_named['bn_b%d%d' % (r, lev)] = bw[lev * R + r], bb[lev * R + r]
    rparams = [ref_named[n] for n in names]
    r1 = torch.autograd.grad(inner_loss(ro), rparams, create_graph=True, allow_unused=True)
    rdot = sum((gi * v).sum() for gi, v in zip(r1, vs) if gi is not None)
    rhv = torch.autograd.grad(rdot, rparams + xr[offset:], allow_unused=True)
    hmax = max(float(r.abs().max()) for r in rhv if r is not None)
    worst = []
    for n, a, r in zip(names + ['x%d' % i for i in range(offset, L)], hv, rhv):
        if r is None:
            assert a is None or float(a.abs().max()) == 0.0, n
            continue
        assert a is not None, n
        worst.append((float((a.cpu() - r).abs().max()) / max(float(r.abs().max()), 1e-3 * hmax), n))
    worst.sort(reverse=True)
    assert worst[0][0] <= 5e-3, worst[:6]
    # ---- the inner step + query loss of infer.py:660-687: outer gradient w.r.t. the parameters and the inner learning rate
    def outer(params_, x_, fwd, lr):
        o = fwd(x_, None)
        gi = torch.autograd.grad(inner_loss(o), params_, create_graph=True, allow_unused=True)
        fast = [p if (gq is None or n.startswith('bn_')) else p - lr * gq for n, p, gq in zip(names, params_, gi)]
        q = fwd(x_, fast)
        return sum((t * w.to(t.device)).sum() for t, w in zip(q, go))
    lr_h = torch.tensor(0.05, device=DEV, requires_grad=True)
    fwd_h = lambda x_, fast: mh([t.detach() for t in x_], fast_weights=fast, level_offset=offset)
    lo = outer(params, xs, fwd_h, lr_h)
    gh = torch.autograd.grad(lo, params + [lr_h], allow_unused=True)
    lr_r = torch.tensor(0.05, requires_grad=True)

    def fwd_r(x_, fast):
        if fast is None:
            return om.meta_head_forward(dw, pw, pb, bw, bb, pred, [t.detach() for t in x_], level_offset=offset)[0]
        f = dict(zip(names, fast))
        return om.meta_head_forward([f['conv_dw%d' % r] for r in range(R)], [f['conv_pw%d' % r] for r in range(R)],
                                    [f['conv_pb%d' % r] for r in range(R)],
                                    [f['bn_w%d%d' % (r, lev)] for lev in range(L) for r in range(R)],
                                    [f['bn_b%d%d' % (r, lev)] for lev in range(L) for r in range(R)],
                                    [f['predict_dw'], f['predict_pw'], f['predict_pb']], [t.detach() for t in x_], level_offset=offset)[0]
    lr_ = outer(rparams, xr, fwd_r, lr_r)
    gr = torch.autograd.grad(lr_, rparams + [lr_r], allow_unused=True)
    assert abs(float(lo) - float(lr_)) <= 2e-4 * max(1.0, abs(float(lr_)))
    gmax = max(float(r.abs().max()) for r in gr if r is not None)
    worst = []
    for n, a, r in zip(names + ['inner_lr'], gh, gr):
        if r is None:
            continue
        assert a is not None, n
        worst.append((float((a.cpu() - r).abs().max()) / max(float(r.abs().max()), 1e-3 * gmax), n))
    worst.sort(reverse=True)
    assert worst[0][0] <= 5e-3, worst[:6]
