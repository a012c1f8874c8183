"""BASELINE.json's configurations at their REAL sizes, and accuracy bounds for the benched bf16 mode.

* d2 / 768 and d4 / 1024: one float32 image against the CPU oracle (head outputs <= 1e-3, the post-processing chain on the same
  logits, soft-NMS on for d4), then the bf16 property checks at the per-GPU bench batch (32 / 8);
* batch invariance at d0 / 640 / B = 64, both dtypes: image i of the batch is bit-equal to the same image run at B = 1 and
  B = 16 (kernel variants are chosen by map size AND grid size: csrc/pw_gemm.hip `small`, csrc/mbconv.hip pick_deep);
* bf16: per-stage error with each stage fed the float32 oracle's input, and detection agreement bf16 vs float32 HIP on the
  bench's own model (`bench.build_model`)."""
import copy

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


def _props(det, cnt, nmax, C, soft, ood):
    """size-independent properties of a DetBenchPredict result (see test_detbench_bench_size_properties)"""
    det, cnt = det.float().cpu(), cnt.cpu()
    assert int(cnt.max()) <= nmax and int(cnt.min()) > 0
    for i in range(det.shape[0]):
        d = det[i, :int(cnt[i])]
        assert torch.isfinite(d).all()
        assert torch.all(d[:-1, 4] >= d[1:, 4]) and torch.all(d[:, 4] > (0.001 if soft else 0.01))
        assert torch.all((d[:, 5] >= 1) & (d[:, 5] <= C)) and torch.all(d[:, 2] >= d[:, 0]) and torch.all(d[:, 3] >= d[:, 1])
        assert torch.all(det[i, int(cnt[i]):] == 0)
        if not soft:
            x1 = torch.max(d[:, None, 0], d[None, :, 0]); y1 = torch.max(d[:, None, 1], d[None, :, 1])
            x2 = torch.min(d[:, None, 2], d[None, :, 2]); y2 = torch.min(d[:, None, 3], d[None, :, 3])
            inter = (x2 - x1).clamp(min=0) * (y2 - y1).clamp(min=0)
            area = (d[:, 2] - d[:, 0]) * (d[:, 3] - d[:, 1])
            iou = inter / (area[:, None] + area[None, :] - inter).clamp(min=1e-9)
            same = (d[:, None, 5] == d[None, :, 5]) & ~torch.eye(d.shape[0], dtype=torch.bool)
            assert (float(iou[same].max()) <= 0.3 + 1e-3) if same.any() else True
    assert torch.isfinite(ood['anchor_energy']).all() and torch.all(-ood['anchor_energy'] >= ood['anchor_max_logit'] - 1e-4)


@pytest.mark.parametrize('name,size,ncls,soft,batch', [('tf_efficientdet_d2', 768, 90, False, 32),      # BASELINE configs[2]: 256 / 8 GPUs
                                                       ('tf_efficientdet_d4', 1024, 90, True, 8)])      # BASELINE configs[3]: soft-NMS
def test_real_size_config(name, size, ncls, soft, batch):
    from ood_object_detection_amd.effdet.bench import DetBenchPredict
    model, cfg, nodes, sd = seeded_model(name, size, ncls, seed=21, cls_bias=-2.0, soft_nms=soft)
    x = torch.from_numpy(seeded_array(21, 'input', (1, 3, size, size)))
    with torch.no_grad():
        cls_r, box_r = om.efficientdet_forward(sd, cfg, x, nodes)
    m = copy.deepcopy(model).to(DEV).float()
    bench = DetBenchPredict(m).to(DEV)
    with torch.no_grad():
        det = bench(x.to(DEV))
    eng = m._engine
    cls_g = [t.float().cpu() for t in eng.head_views(eng.cls_all, ncls)]
    box_g = [t.float().cpu() for t in eng.head_views(eng.box_all, 4)]
    # (i) the network at its real size, float32: head outputs against the oracle (north star: 1e-3 abs)
    err_c = max(_linf(a, r) for a, r in zip(cls_g, cls_r))
    err_b = max(_linf(a, r) for a, r in zip(box_g, box_r))
    print('%s %d px fp32: class logits L-inf %.2e, box outputs L-inf %.2e' % (name, size, err_c, err_b))
    assert err_c <= 1e-3 and err_b <= 1e-3
    e_ref, m_ref = om.ood_scores(cls_r, ncls)
    assert _linf(m.ood_energy, e_ref) <= 1e-3 and _linf(m.ood_max_logit, m_ref) <= 1e-3
    # (ii) post-processing at the real anchor count against the oracle fed the SAME logits
    anchors = op.anchor_boxes(cfg.min_level, cfg.max_level, cfg.num_scales, cfg.aspect_ratios, cfg.anchor_scale, (size, size))
    assert anchors.shape[0] == {768: 110484, 1024: 196416}[size]
    c, b, idx, cl = op.post_process(cls_g, box_g, 5, ncls, 5000)
    ref, src = op.generate_detections(c[0], b[0], anchors, idx[0], cl[0], None, torch.tensor(size), 100, soft, return_aux=True)
    n = int(bench.last_count[0])
    assert n == ref.shape[0] and n > 0
    got = det[0, :n].cpu()
    assert torch.equal(got[:, 5], ref[:, 5])
    assert float((got[:, 4] - ref[:, 4]).abs().max()) <= 1e-5 and float((got[:, :4] - ref[:, :4]).abs().max()) <= 1e-3
    e_same, m_same = om.ood_scores(cls_g, ncls)
    a_idx = idx[0][src]
    assert float((bench.last_ood['energy'][0, :n].cpu() - e_same[0][a_idx]).abs().max()) <= 1e-4
    del bench, m
    # (iii) the bench configuration: bf16 at the per-GPU batch, size-independent properties + image 0 equals the B = 1 run
    mb = copy.deepcopy(model).to(DEV).to(torch.bfloat16)
    xb = torch.from_numpy(seeded_array(22, 'batch', (batch, 3, size, size))).to(DEV).to(torch.bfloat16)
    benchb = DetBenchPredict(mb).to(DEV)
    with torch.no_grad():
        detb = benchb(xb)
        cntb, oodb = benchb.last_count.clone(), {k: v.clone() for k, v in benchb.last_ood.items()}
        _props(detb, cntb, 100, ncls, soft, oodb)
        det1 = DetBenchPredict(mb, streams=1).to(DEV)(xb[:1])
    assert torch.equal(det1[0], detb[0])
    # bf16 vs float32 on the same image: scores of the strongest detections agree (calibrated network, logits O(1))
    with torch.no_grad():
        cls_b, box_b = mb(x.to(DEV).to(torch.bfloat16))
    rel = max(float((a.float().cpu() - r).abs().max()) / max(1.0, float(r.abs().max())) for a, r in zip(list(cls_b) + list(box_b), list(cls_r) + list(box_r)))
    print('%s %d px bf16 vs fp32 oracle: head outputs, worst L-inf / max|ref| = %.3f' % (name, size, rel))
    assert rel <= 0.1                         # measured on MI355X (round 2): 0.053 (d2 / 768), 0.021 (d4 / 1024)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_batch_invariance_d0_640_b64(dtype):
    """the headline configuration itself (d0 / 640 / batch 64): every image of the batch must be bit-equal to the same image
    run alone and in a batch of 16 - catches every grid-size-dependent kernel variant"""
    import bench as B
    model = B.build_model('tf_efficientdet_d0', 640, 90).to(DEV).to(dtype)
    x = torch.randn(64, 3, 640, 640, device=DEV, generator=torch.Generator(device=DEV).manual_seed(7)).to(dtype)
    with torch.no_grad():
        c64, b64 = model(x)
        c64 = [t.clone() for t in c64]; b64 = [t.clone() for t in b64]
        e64 = model.ood_energy.clone()
        for i in (0, 37, 63):
            c1, b1 = model(x[i:i + 1])
            for a, r in zip(list(c1) + list(b1), c64 + b64):
                assert torch.equal(a[0], r[i]), 'image %d differs between B=1 and B=64' % i
            assert torch.equal(model.ood_energy[0], e64[i])
        c16, b16 = model(x[16:32])
        for a, r in zip(list(c16) + list(b16), c64 + b64):
            assert torch.equal(a, r[16:32]), 'B=16 differs from B=64'
        assert torch.equal(model.ood_energy, e64[16:32])
    # detections of the full DetBenchPredict path as well (two concurrent half-batches at B = 64)
    from ood_object_detection_amd.effdet.bench import DetBenchPredict
    with torch.no_grad():
        d64 = DetBenchPredict(model).to(DEV)(x).clone()
        d1 = DetBenchPredict(model, streams=1).to(DEV)(x[37:38])
    assert torch.equal(d1[0], d64[37])


def test_bf16_per_stage_error_bounds():
    """bf16 (the benched mode) with a bound that can fail: each stage (backbone, BiFPN, heads) is fed the float32 ORACLE's
    input for that stage, rounded to bf16, and its output must stay within a few bf16 ulps-of-max of the oracle's output -
    error does not compound across stages in this test, so a broken kernel variant shows up as a stage far outside its band"""
    model, cfg, nodes, sd = seeded_model('tf_efficientdet_d0', 256, 90, seed=3)
    x = torch.from_numpy(seeded_array(3, 'input', (2, 3, 256, 256)))
    with torch.no_grad():
        feats, activs = om.efficientdet_forward(sd, cfg, x, nodes, mode='fpn')
        cls_r, box_r = om.efficientdet_forward(sd, cfg, x, nodes)
    m = copy.deepcopy(model).to(DEV).to(torch.bfloat16)
    bf = lambda ts: [t.to(DEV).to(torch.bfloat16) for t in ts]

    def worst(a, r):
        return float((a.float().cpu() - r).abs().max()) / max(1e-6, float(r.abs().max()))

    def rms(a, r):
        return float((a.float().cpu() - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt().clamp(min=1e-6))
    with torch.no_grad():
        f_g = [t.clone() for t in m(x.to(DEV).to(torch.bfloat16), mode='bb')]
        a_g = [t.clone() for t in m(bf(feats), mode='only_fpn')]
        c_g, b_g = m(bf(activs), mode='head')
    rows = [('backbone.P3 (5 blocks deep)', f_g[0], feats[0], 0.04, 0.03), ('backbone.P4 (11 blocks)', f_g[1], feats[1], 0.06, 0.05),
            ('backbone.P5 (16 blocks)', f_g[2], feats[2], 0.085, 0.075)]
    rows += [('BiFPN level %d' % i, a_g[i], activs[i], 0.05, 0.03) for i in range(5)]
    rows += [('class head level %d' % i, c_g[i], cls_r[i], 0.02, 0.015) for i in range(5)]
    rows += [('box head level %d' % i, b_g[i], box_r[i], 0.02, 0.015) for i in range(5)]
    bad = []
    for name, a, r, lim_inf, lim_rms in rows:
        e, q = worst(a, r), rms(a, r)
        print('bf16 stage error  %-28s L-inf/max|ref| %.4f (<= %.3f)   rel-rms %.4f (<= %.3f)' % (name, e, lim_inf, q, lim_rms))
        if e > lim_inf or q > lim_rms:
            bad.append(name)
    # bands = 1.5 x the values measured on MI355X (round 2) for one stage of bf16 storage (2^-9 per rounding) on this seeded
    # network, which amplifies perturbations by ~30x over the backbone's depth; a wrong tile / variant is off by O(1)
    assert not bad, bad


def test_bf16_detection_agreement_on_the_bench_model():
    """detection-level accuracy of the benched mode: DetBenchPredict in bf16 against the float32 HIP path on bench.build_model's
    weights at d0 / 640 (what bench.py's `parity_bf16` block reports).  Discrete decisions (top-k membership, NMS) may flip for
    near-ties, so the bound is on the matched fraction and on the error of the matched pairs."""
    import bench as B
    from ood_object_detection_amd.effdet.bench import DetBenchPredict
    model = B.build_model('tf_efficientdet_d0', 640, 90)
    x = torch.randn(4, 3, 640, 640, generator=torch.Generator().manual_seed(5))
    res = B.parity_bf16(model, x, DEV)
    print('bf16 vs f32 detections on the bench model:', res)
    # this network's logits span +-0.15 (every score ~ 0.5): top-k and NMS decide between near-ties, so only about half of the
    # kept detections coincide; the ones that do, and ALL of them on equal candidates, must agree closely
    # bounds = 1.5 x the values measured on MI355X in round 3 (matched 0.59, boxes 0.46 px, scores 2.1e-4; kept in
    # profiles/r03_*_bench_line.json as `parity_bf16`): a regression guard derived from measurements, tight enough to fail on a
    # wrong tile or a dropped rounding
    assert res['matched_frac'] >= 0.45
    assert res['scores_linf'] <= 3.2e-4 and res['boxes_linf_px'] <= 0.70
    assert res['same_candidates']['scores_linf'] <= 3.2e-4 and res['same_candidates']['boxes_linf_px'] <= 0.70
    # the mixed mode (bfloat16 backbone, float32 BiFPN + heads): measured 0.98 matched, 0.007 px, 1e-6
    mix = B.parity_bf16(model, x, DEV, candidate='mixed')
    print('mixed vs f32 detections on the bench model:', mix)
    assert mix['matched_frac'] >= 0.9 and mix['same_candidates']['boxes_linf_px'] <= 0.02 and mix['same_candidates']['scores_linf'] <= 1e-5


def test_bf16_detection_agreement_calibrated_network():
    """the same measurement on the BN-calibrated seeded network (logits O(1), scores spread over (0, 1)): near-ties are rare, so
    most detections coincide"""
    import bench as B
    model, cfg, nodes, sd = seeded_model('tf_efficientdet_d0', 512, 90, seed=11, cls_bias=-2.0)
    x = torch.from_numpy(seeded_array(12, 'input', (4, 3, 512, 512)))
    res = B.parity_bf16(model, x, DEV)
    print('bf16 vs f32 detections on the calibrated network:', res)
    # measured in round 3 (bench line `parity_bf16_calibrated`): matched 0.905, boxes 1.25 px, scores 0.0099; bounds = 1.5 x
    assert res['matched_frac'] >= 0.8
    assert res['same_candidates']['scores_linf'] <= 0.015 and res['same_candidates']['boxes_linf_px'] <= 1.9
