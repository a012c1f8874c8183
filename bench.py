#!/usr/bin/env python
"""Headline benchmark: images/sec of tf_efficientdet_d0 at 640x640, bf16, batch 64 per GPU, through
DetBenchPredict.forward (backbone -> BiFPN -> heads + per-anchor OOD energy/max-logit -> top-k ->
decode -> NMS), inputs resident in HBM, synthetic data, seeded random-init weights.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; images are sharded (weak scaling: 64 per GPU), no collective on the data path.
A step is one full DetBenchPredict.forward over the rank's 64-image batch.  Three batches are kept in flight: consecutive
steps alternate between three engine instances (own activation buffers and hipGraph, shared weights) on three streams, so
the latency-bound tail of one step (top-k, NMS, small BiFPN levels, SE gates) overlaps the wide kernels of the next - the
way a serving loop would pipeline requests.  Every timed step runs to completion inside the timed region
(`--in-flight 1 --sub-batches 0` restores one batch at a time as two concurrent half-batches: 11.5 k instead of 12.7 k img/s).
Rank 0 prints ONE JSON line (contract in the round prompt) that also carries
  roofline      dominant kernel family: algorithmic HBM bytes / HIP-event time on the launch stream
  cpu_baseline  the CPU oracle (oracle/, a port of the reference's PyTorch path) timed on this host
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT,):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_FLOPS = 2.5e15       # MI355X_MICROARCH.md: dense bf16 MFMA peak (the fp32 parity mode would be lower; the bench runs bf16)


def build_model(name, image, num_classes, seed=0):
    """Reference init (_init_weight + timm-style backbone init) with the BN running stats and BiFPN edge
    weights randomised so that nothing folds to identity; class-predict bias 0 ('trained-like': thousands of
    candidates pass the 0.01 score threshold, so top-k / NMS do real work)."""
    from ood_object_detection_amd.effdet.factory import create_model
    torch.manual_seed(seed)
    model = create_model(name, num_classes=num_classes, image_size=(image, image)).eval()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, m in model.named_modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
            if hasattr(m, 'edge_weights') and m.edge_weights is not None:
                m.edge_weights.copy_(torch.rand(m.edge_weights.shape, generator=g) * 2.0)
        model.class_net.predict.conv_pw.bias.zero_()
    return model


def host_cores(detail=False):
    """CPU share of this process: min(affinity mask, cgroup cpu quota) - a 1-GPU box gets ~16 of the host's cores, and asking
    torch for more threads than that share oversubscribes badly.  EFFDET_CPU_THREADS (unset by default) overrides the count.
    detail=True returns what the choice was made from, for the bench line."""
    cpu_count = os.cpu_count() or 1
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else cpu_count
    quota = None
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    quota = max(1, int(math.ceil(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                    quota = max(1, int(math.ceil(q / per)))
            break
        except Exception:
            continue
    n = min(affinity, quota) if quota else affinity
    override = os.environ.get('EFFDET_CPU_THREADS')
    if override:
        n = max(1, int(override))
    n = max(1, n)
    if detail:
        return n, {'os_cpu_count': cpu_count, 'affinity': affinity, 'cgroup_quota_cpus': quota, 'threads_override': int(override) if override else None}
    return n


def cpu_baseline(model_cpu_sd, cfg, image, num_classes, budget_s=20.0, gpu_check=None):
    """The oracle's full path (forward + top-k + decode + hard NMS + OOD) on the host cores.
    gpu_check(x) -> (cls_outs, box_outs, energy) from the HIP path in float32 on the same image: the L-inf
    against the oracle's outputs is reported next to the baseline (north star: <= 1e-3 abs)."""
    from oracle import model as om
    from oracle import postprocess as op
    from ood_object_detection_amd.effdet.config import get_fpn_config
    threads, core_info = host_cores(detail=True)
    torch.set_num_threads(threads)
    nodes = get_fpn_config(cfg.fpn_name, cfg.min_level, cfg.max_level).nodes
    anchors = op.anchor_boxes(cfg.min_level, cfg.max_level, cfg.num_scales, cfg.aspect_ratios, cfg.anchor_scale, (image, image))
    B = 1
    x = torch.randn(B, 3, image, image, generator=torch.Generator().manual_seed(5))
    split = {'forward_s': 0.0, 'postprocess_s': 0.0}       # where the oracle's time goes (network vs top-k / decode / NMS loops)

    def step(xb=None):
        xb = x if xb is None else xb
        with torch.no_grad():
            t0 = time.time()
            cls_o, box_o = om.efficientdet_forward(model_cpu_sd, cfg, xb, nodes)
            om.ood_scores(cls_o, num_classes)
            t1 = time.time()
            c, b, idx, cl = op.post_process(cls_o, box_o, cfg.num_levels, num_classes, cfg.max_detection_points)
            for i in range(xb.shape[0]):
                op.generate_detections(c[i], b[i], anchors, idx[i], cl[i], None, torch.tensor(image), cfg.max_det_per_image, False)
            split['forward_s'] += t1 - t0
            split['postprocess_s'] += time.time() - t1

    parity = None
    if gpu_check is not None:
        with torch.no_grad():
            cls_o, box_o = om.efficientdet_forward(model_cpu_sd, cfg, x, nodes)
            energy, _ = om.ood_scores(cls_o, num_classes)
            gc, gb, ge = gpu_check(x)
        parity = {'dtype': 'f32', 'images': B,
                  'class_logits_linf': float(max((a - b.cpu()).abs().max() for a, b in zip(cls_o, gc))),
                  'box_outputs_linf': float(max((a - b.cpu()).abs().max() for a, b in zip(box_o, gb))),
                  'ood_energy_linf': float((energy - ge.cpu()).abs().max())}

    def timed(xb, budget):
        t0 = time.time()
        step(xb)                              # first pass: warm-up, and the measurement itself if it is very slow
        first = time.time() - t0
        n, dt = 1, first
        if first < budget / 2:
            split['forward_s'] = split['postprocess_s'] = 0.0
            n, t1 = 0, time.time()
            while (time.time() - t1) < (budget - first) and n < 50:
                step(xb)
                n += 1
            dt = time.time() - t1
        per_image = {k: round(v / (n * xb.shape[0]), 4) for k, v in split.items()}
        split['forward_s'] = split['postprocess_s'] = 0.0
        return xb.shape[0] * n / dt, n, per_image
    # batch 1 and batch 8 (the larger batch gives the host's GEMMs more rows per call)
    r1, n1, sp1 = timed(x, budget_s * 0.6)
    x8 = torch.randn(8, 3, image, image, generator=torch.Generator().manual_seed(6))
    r8, n8, sp8 = timed(x8, budget_s * 0.4)
    best = sp8 if r8 >= r1 else sp1
    out = {'value': round(max(r1, r8), 3), 'unit': 'images/sec', 'cores': torch.get_num_threads(), 'kind': 'port',
           'sample': '%d x batch-1 and %d x batch-8 fp32 oracle passes of the same %d px workload (forward+OOD+top-k+decode+hard NMS); '
                     'value = the faster of the two' % (n1, n8, image),
           'batch1_images_per_sec': round(r1, 3), 'batch8_images_per_sec': round(r8, 3),
           # seconds per image of the faster configuration: the network (torch CPU convolutions) vs the oracle's post-processing
           # (stable sort of N*C logits + the Python NMS loop) - the latter is a checker, not an optimised CPU implementation
           'forward_s_per_image': best['forward_s'], 'postprocess_s_per_image': best['postprocess_s'],
           'postprocess_share': round(best['postprocess_s'] / max(1e-9, best['forward_s'] + best['postprocess_s']), 3),
           'forward_only_images_per_sec': round(1.0 / max(1e-9, best['forward_s']), 3),
           'host': dict(core_info, threads_used=torch.get_num_threads())}
    if parity is not None:
        out['parity_vs_hip_f32'] = parity
    return out


def detection_agreement(det_a, cnt_a, anc_a, det_b, cnt_b, anc_b):
    """Agreement of two DetBenchPredict results ([B, max_det, 6] = x1, y1, x2, y2, score, class; valid rows per image in cnt;
    anc = the anchor index every kept detection came from, `last_ood['anchor_index']`).  Two detections are THE SAME
    detection when they come from the same anchor with the same class.  Returns the fraction of A's detections that B also
    has and, over those pairs, the L-inf of box coordinates (pixels) and of scores - the detection-level counterpart of
    BASELINE.json's 'box/score L-inf vs ref'.  Unmatched detections are near-ties of the discrete steps (top-k, NMS)."""
    det_a, det_b, anc_a, anc_b = det_a.float().cpu(), det_b.float().cpu(), anc_a.cpu(), anc_b.cpu()
    n_ref = n_match = 0
    box_linf = score_linf = 0.0
    for i in range(det_a.shape[0]):
        na, nb = int(cnt_a[i]), int(cnt_b[i])
        kb = {(int(anc_b[i, j]), int(det_b[i, j, 5])): j for j in range(nb)}
        n_ref += na
        for r in range(na):
            j = kb.get((int(anc_a[i, r]), int(det_a[i, r, 5])))
            if j is None:
                continue
            n_match += 1
            box_linf = max(box_linf, float((det_b[i, j, :4] - det_a[i, r, :4]).abs().max()))
            score_linf = max(score_linf, float(abs(det_b[i, j, 4] - det_a[i, r, 4])))
    return {'matched_frac': round(n_match / max(1, n_ref), 4), 'boxes_linf_px': round(box_linf, 4), 'scores_linf': round(score_linf, 6),
            'detections_ref': n_ref}


def parity_bf16(model_f32_cpu, x_cpu, dev, soft_nms=False, candidate='bf16'):
    """A reduced-precision mode against the float32 HIP path, same weights, same images, whole DetBenchPredict: detection
    agreement plus the L-inf of the head outputs and of the per-anchor OOD energy.  candidate = 'bf16' (the benched mode) or
    'mixed' (bfloat16 backbone, float32 BiFPN + heads: serving.MixedPrecisionEfficientDet)."""
    import copy
    from ood_object_detection_amd.effdet.bench import DetBenchPredict
    from ood_object_detection_amd.serving import MixedPrecisionEfficientDet
    out = {}
    res = {}
    for tag in ('f32', candidate):
        m = copy.deepcopy(model_f32_cpu).to(dev)
        dt = torch.float32
        if tag == 'bf16':
            m, dt = m.to(torch.bfloat16), torch.bfloat16
        elif tag == 'mixed':
            m, dt = MixedPrecisionEfficientDet(m), torch.bfloat16
        m.config.soft_nms = bool(soft_nms)
        b = DetBenchPredict(m, streams=1).to(dev)
        with torch.no_grad():
            det = b(x_cpu.to(dev).to(dt))
        eng = m._engine
        res[tag] = (det.float().cpu(), b.last_count.cpu(), eng.cls_all.float().cpu(), eng.box_all.float().cpu(), m.ood_energy.float().cpu(),
                    b.last_ood['anchor_index'].cpu())
        anchors = b.anchors.boxes.float().cpu()
        del b, m
    out.update(detection_agreement(res['f32'][0], res['f32'][1], res['f32'][5], res[candidate][0], res[candidate][1], res[candidate][5]))
    # the same comparison with the discrete steps taken out: for every detection the float32 path kept, what the candidate's head
    # outputs give for that SAME (anchor, class) - score = sigmoid(logit), box = decode (effdet/anchors.py:51-85) - so every
    # reference detection is covered whether or not top-k / NMS made the same choice
    det32, cnt32, cls32, box32, _, anc32 = res['f32']
    cls16, box16 = res[candidate][2], res[candidate][3]

    def decode(rel, a):
        ya, xa, ha, wa = (a[:, 0] + a[:, 2]) / 2, (a[:, 1] + a[:, 3]) / 2, a[:, 2] - a[:, 0], a[:, 3] - a[:, 1]
        w, h = torch.exp(rel[:, 3]) * wa, torch.exp(rel[:, 2]) * ha
        yc, xc = rel[:, 0] * ha + ya, rel[:, 1] * wa + xa
        return torch.stack([xc - w / 2, yc - h / 2, xc + w / 2, yc + h / 2], 1)
    sb = ss = 0.0
    for i in range(det32.shape[0]):
        n = int(cnt32[i])
        if n == 0:
            continue
        a_idx, c_idx = anc32[i, :n], det32[i, :n, 5].long() - 1
        ss = max(ss, float((torch.sigmoid(cls32[i, a_idx, c_idx]) - torch.sigmoid(cls16[i, a_idx, c_idx])).abs().max()))
        sb = max(sb, float((decode(box32[i, a_idx], anchors[a_idx]) - decode(box16[i, a_idx], anchors[a_idx])).abs().max()))
    out['same_candidates'] = {'boxes_linf_px': round(sb, 4), 'scores_linf': round(ss, 6)}
    out['images'] = int(x_cpu.shape[0])
    out['class_logits_linf'] = round(float((res['f32'][2] - res[candidate][2]).abs().max()), 5)
    out['box_outputs_linf'] = round(float((res['f32'][3] - res[candidate][3]).abs().max()), 5)
    out['ood_energy_linf'] = round(float((res['f32'][4] - res[candidate][4]).abs().max()), 5)
    out['class_logits_absmax'] = round(float(res['f32'][2].abs().max()), 4)
    out['score_spread'] = [round(float(v), 4) for v in torch.sigmoid(res['f32'][2]).flatten()[::97].quantile(torch.tensor([0.01, 0.5, 0.99]))]
    out['reference'] = 'float32 HIP path (itself checked against the CPU oracle: parity_vs_hip_f32)'
    return out


def parity_accurate_vs_oracle(model_f32_cpu, x_cpu, dev, image, num_classes, soft_nms=False):
    """compute_mode='accurate' (two-term bf16) against the CPU ORACLE (float32 PyTorch restatement of the reference) on the same
    weights and images: L-inf of class logits, box regressions, OOD energy / max-logit, and - for every detection the HIP path kept -
    of its score and decoded box against the oracle's values for the SAME (anchor, class).  Uses oracle/: cpu_baseline leg only."""
    import copy
    from oracle import model as om
    from ood_object_detection_amd.effdet.bench import DetBenchPredict
    from ood_object_detection_amd.effdet.config import get_fpn_config
    cfg = model_f32_cpu.config
    if x_cpu is None:
        x_cpu = torch.randn(2, 3, image, image, generator=torch.Generator().manual_seed(5))
    x_cpu = x_cpu[:2]
    B, C = x_cpu.shape[0], num_classes
    sd = {k: v.detach().clone().float() for k, v in model_f32_cpu.state_dict().items()}
    nodes = get_fpn_config(cfg.fpn_name, cfg.min_level, cfg.max_level).nodes
    with torch.no_grad():
        cls_r, box_r = om.efficientdet_forward(sd, cfg, x_cpu, nodes)
        e_ref, m_ref = om.ood_scores(cls_r, C)
    m = copy.deepcopy(model_f32_cpu).to(dev).float()
    m.compute_mode = 'accurate'
    m.config.soft_nms = bool(soft_nms)
    b = DetBenchPredict(m, streams=1).to(dev)
    with torch.no_grad():
        det = b(x_cpu.to(dev)).float().cpu()
    eng = m._engine
    cls_g = eng.cls_all.float().cpu()
    box_g = eng.box_all.float().cpu()
    cls_ref = torch.cat([r.permute(0, 2, 3, 1).reshape(B, -1, C) for r in cls_r], 1)
    box_ref = torch.cat([r.permute(0, 2, 3, 1).reshape(B, -1, 4) for r in box_r], 1)
    anchors = b.anchors.boxes.float().cpu()
    anc = b.last_ood['anchor_index'].cpu()

    def decode(rel, a):
        ya, xa, ha, wa = (a[:, 0] + a[:, 2]) / 2, (a[:, 1] + a[:, 3]) / 2, a[:, 2] - a[:, 0], a[:, 3] - a[:, 1]
        w, h = torch.exp(rel[:, 3]) * wa, torch.exp(rel[:, 2]) * ha
        yc, xc = rel[:, 0] * ha + ya, rel[:, 1] * wa + xa
        return torch.stack([xc - w / 2, yc - h / 2, xc + w / 2, yc + h / 2], 1)
    ss = sb = 0.0
    ndet = 0
    for i in range(B):
        n = int(b.last_count[i])
        ndet += n
        if n == 0:
            continue
        a_idx, c_idx = anc[i, :n], det[i, :n, 5].long() - 1
        if not soft_nms:                                     # (soft-NMS rescales the scores it keeps)
            ss = max(ss, float((torch.sigmoid(cls_ref[i, a_idx, c_idx]) - det[i, :n, 4]).abs().max()))
        sb = max(sb, float((decode(box_ref[i, a_idx], anchors[a_idx]) - det[i, :n, :4]).abs().max()))
    return {'reference': 'CPU oracle (float32 PyTorch restatement of the reference)', 'images': B, 'image_px': int(x_cpu.shape[2]),
            'class_logits_linf': round(float((cls_g - cls_ref).abs().max()), 7), 'box_outputs_linf': round(float((box_g - box_ref).abs().max()), 7),
            'ood_energy_linf': round(float((m.ood_energy.cpu() - e_ref).abs().max()), 7),
            'ood_max_logit_linf': round(float((m.ood_max_logit.cpu() - m_ref).abs().max()), 7),
            'same_candidates': {'scores_linf': round(ss, 7), 'boxes_linf_px': round(sb, 5), 'detections': ndet},
            'class_logits_absmax': round(float(cls_ref.abs().max()), 4)}


def calibrated_model(image=512, num_classes=90, seed=11):
    """The BN-calibrated seeded network of the parity tests (tests/_models.py): every BatchNorm's running statistics are what the
    layer actually sees on seeded images (set by one pass of the CPU oracle), class logits O(1), scores spread over (0, 1) - the
    well-conditioned counterpart of `build_model`'s reference init, whose logits all lie within +-0.15.  Uses oracle/: called
    from the cpu_baseline leg only."""
    tdir = os.path.join(ROOT, 'tests')
    if tdir not in sys.path:
        sys.path.insert(0, tdir)
    from _models import seeded_model
    from _seeded import seeded_array
    model, cfg, nodes, sd = seeded_model('tf_efficientdet_d0', image, num_classes, seed=seed, cls_bias=-2.0)
    x = torch.from_numpy(seeded_array(seed + 1, 'input', (4, 3, image, image)))
    return model, x


def kernels_sha16():
    """content hash of the inference kernel sources: ties committed PMC traffic files to the kernels they were measured on
    (the training-only translation units do not run in this bench and are left out)"""
    import glob, hashlib
    h = hashlib.sha256()
    root = os.path.join(ROOT, 'ood_object_detection_amd', 'csrc')
    skip = ('train_ops.hip', 'train_net.hip', 'train_levels.hip', 'train_fpn.hip', 'evaluation.hip')
    for f in sorted(glob.glob(os.path.join(root, '*.hip')) + glob.glob(os.path.join(root, '*.h'))):
        if os.path.basename(f) in skip:
            continue
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def auroc_surrogate(model, image, dev, dtype, n_img=32, bs=8, planted=3, seed=7):
    """BASELINE config 4 surrogate: there is no COCO / OpenImages-unseen data and no trained checkpoint here, so
    in-distribution = images whose class logits carry a few PLANTED high-confidence anchors (logit +6 on one class of `planted`
    random anchors per image - what a trained detector produces on an object it knows), OOD = the same network on noise images
    with the logits as they come out.  Both go through the real pipeline (backbone -> BiFPN -> heads -> per-anchor energy in the
    class-head epilogue -> image score max_a(-energy_a) -> pair-counting AUROC, all HIP kernels); for the planted anchors the
    energy is updated from the stored logits with the same -logsumexp.  Synthetic by construction; reported as such."""
    from ood_object_detection_amd import ood
    g = torch.Generator(device=dev).manual_seed(seed)
    scores = {'in': [], 'ood': []}
    C = model.config.num_classes
    with torch.no_grad():
        for kind in ('in', 'ood'):
            for _ in range(n_img // bs):
                x = torch.randn(bs, 3, image, image, device=dev, generator=g).to(dtype)
                model(x)
                eng = model._engine
                energy = eng.ood_energy.clone()                       # [bs, N] float32 from the class-head epilogue
                if kind == 'in':
                    N = energy.shape[1]
                    a = torch.randint(0, N, (bs, planted), device=dev, generator=g)
                    c = torch.randint(0, C, (bs, planted), device=dev, generator=g)
                    rows = eng.cls_all.float()[torch.arange(bs, device=dev)[:, None], a]          # [bs, planted, C] stored logits
                    rows[torch.arange(bs, device=dev)[:, None], torch.arange(planted, device=dev)[None, :], c] = 6.0
                    energy[torch.arange(bs, device=dev)[:, None], a] = -torch.logsumexp(rows, dim=2)
                scores[kind].append(ood.image_scores(energy.contiguous()).clone())
    s_in, s_ood = torch.cat(scores['in']), torch.cat(scores['ood'])
    return {'auroc_in_dist_positive': round(ood.auroc(s_in, s_ood), 4), 'score': 'max_a(-energy_a)',
            'mean_score_in': round(float(s_in.mean()), 4), 'mean_score_ood': round(float(s_ood.mean()), 4),
            'images': '%d in-dist (planted: %d anchors per image at logit +6) + %d OOD (noise)' % (n_img, planted, n_img), 'data': 'synthetic'}


def timed_steps(model, x, dev, nfl, sub, steps, warmup, use_graph, barrier):
    """W warm-up + K timed DetBenchPredict.forward steps with `nfl` batches in flight (own engine instance, input and stream
    per slot, shared weights).  Returns (elapsed seconds, launch mode, first bench)."""
    import copy
    from ood_object_detection_amd.effdet.bench import DetBenchPredict
    benches = [DetBenchPredict(model if i == 0 else copy.copy(model), streams=sub).to(dev) for i in range(nfl)]
    xs = [x] + [x.clone() for _ in range(nfl - 1)]
    streams = [torch.cuda.Stream(dev) for _ in range(nfl)]
    with torch.no_grad():
        for _ in range(max(1, min(warmup, 3))):
            for b_, x_ in zip(benches, xs):
                b_(x_)
        torch.cuda.synchronize(dev)
        launch, graphs = 'eager', None
        if use_graph:
            try:
                graphs = []
                for b_, x_, s_ in zip(benches, xs, streams):
                    s_.wait_stream(torch.cuda.current_stream(dev))
                    with torch.cuda.stream(s_):
                        b_(x_)
                    torch.cuda.current_stream(dev).wait_stream(s_)
                    g_ = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g_, stream=s_):
                        b_(x_)
                    graphs.append(g_)
                for g_, s_ in zip(graphs, streams):
                    with torch.cuda.stream(s_):
                        g_.replay()
                torch.cuda.synchronize(dev)
                launch = 'hipgraph'
            except Exception as e:          # launch-mode choice only: the same HIP kernels run either way
                sys.stderr.write('graph capture unavailable (%s); launching eagerly\n' % (e,))
                graphs = None
                torch.cuda.synchronize(dev)

        def step(i):
            k = i % nfl
            with torch.cuda.stream(streams[k]):
                if graphs is not None:
                    graphs[k].replay()
                else:
                    benches[k](xs[k])
        for i in range(warmup):
            step(i)
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        barrier()
        elapsed = time.perf_counter() - t0
    return elapsed, launch, benches[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)      # 0.45 s of timed work: 20 steps (90 ms) scatter by +-2 % run to run
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--model', default='tf_efficientdet_d0')
    ap.add_argument('--image', type=int, default=640)
    ap.add_argument('--batch', type=int, default=64, help='images per GPU')
    ap.add_argument('--classes', type=int, default=90)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32', 'accurate'],
                    help="accurate: float32 weights, compute_mode='accurate' (two-term bf16 values, three matrix-core products per multiply)")
    ap.add_argument('--no-graph', action='store_true', help='launch eagerly instead of replaying one hipGraph')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the in-flight-1 / float32 / parity_bf16 side measurements')
    ap.add_argument('--soft-nms', action='store_true')
    ap.add_argument('--auroc-surrogate', action='store_true', help='BASELINE config 4: add the synthetic in-dist / OOD AUROC block')
    ap.add_argument('--profile-out', default='')
    ap.add_argument('--in-flight', type=int, default=3,
                    help='batches in flight: consecutive steps alternate between this many engine instances / streams, so the '
                         'latency-bound tail of one step (top-k, NMS, small BiFPN levels) overlaps the next step\'s wide kernels')
    ap.add_argument('--sub-batches', type=int, default=1, help='concurrent sub-batches inside one forward (0: DetBenchPredict default = 2 for B >= 16)')
    args = ap.parse_args()

    # --gpus N is the contract: with WORLD_SIZE set (torch.distributed.run started us) it must equal N; with WORLD_SIZE unset
    # and N > 1 this process is only the PARENT - before any torch.cuda call it starts N fresh rank processes of this same
    # command (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), waits and exits with the worst child code.
    from ood_object_detection_amd.sharding import launch_ranks, resolve_world
    try:
        rank, local_rank, world, must_launch = resolve_world(args.gpus)
    except ValueError as e:
        raise SystemExit('bench.py: %s' % e)
    if must_launch:
        worker = os.environ.get('EFFDET_BENCH_WORKER', os.path.abspath(__file__))     # tests substitute a GPU-free worker
        sys.exit(launch_ranks(world, [sys.executable, worker] + sys.argv[1:]))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the product path has no CPU fallback')
    # EFFDET_DIST_BACKEND=gloo rehearses the multi-rank control flow on a box with fewer GPUs than ranks
    # (ranks then share devices round-robin); the real runs use RCCL, one rank per GPU.
    backend = os.environ.get('EFFDET_DIST_BACKEND', 'nccl')
    dev = torch.device('cuda', local_rank if backend == 'nccl' else local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from ood_object_detection_amd.effdet.bench import DetBenchPredict
    dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    model = build_model(args.model, args.image, args.classes)
    cfg = model.config
    cfg.soft_nms = bool(args.soft_nms)
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline       # the CPU baseline is an N=1 item
    sd_cpu = {k: v.clone().float() for k, v in model.state_dict().items()} if want_cpu else None
    model = model.to(dev).to(dtype)
    if args.dtype == 'accurate':
        model.compute_mode = 'accurate'
    sub = args.sub_batches or None
    B = args.batch
    x = (torch.randn(B, 3, args.image, args.image, device=dev, generator=torch.Generator(device=dev).manual_seed(100 + rank))).to(dtype)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    import copy
    nfl = max(1, args.in_flight)
    # every in-flight slot has its own engine (activation buffers, launch plan) on shallow copies of the model (shared weights),
    # its own input batch and its own stream; each step still is one full DetBenchPredict.forward over B images
    elapsed, launch, bench = timed_steps(model, x, dev, nfl, sub, args.steps, args.warmup, not args.no_graph, barrier)

    rate_local = B * args.steps / elapsed                   # this rank's own images / s over its own clock
    per_rank, ranks_seen = [round(rate_local, 1)], 1
    if dist is not None:
        cdev = dev if backend == 'nccl' else 'cpu'
        t = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        ones = torch.ones(1, device=cdev, dtype=torch.float64)
        dist.all_reduce(ones)                               # ranks that actually took part in a collective (RCCL when backend = nccl)
        ranks_seen = int(round(float(ones.item())))
        rates = torch.zeros(world, device=cdev, dtype=torch.float64)
        rates[rank] = rate_local
        dist.all_reduce(rates)
        per_rank = [round(float(v), 1) for v in rates.tolist()]
        if ranks_seen != world or dist.get_world_size() != world:
            raise SystemExit('bench.py: %d ranks answered the all-reduce, --gpus %d' % (ranks_seen, world))
    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    ms_per_step = 1e3 * elapsed / args.steps
    value = world * B * args.steps / elapsed
    counts = bench.last_count.float()
    # ---- roofline of the dominant kernel family (HIP events on the launch stream) -------------------
    # The timed step may run as concurrent half-batches (DetBenchPredict streams); the per-launch table is taken
    # on a full-batch launch plan of the same weights, one launch at a time, so that bytes and times per launch
    # refer to the same thing as the rocprofv3 / PMC summaries under profiles/.
    pmodel = copy.copy(model)
    pbench = DetBenchPredict(pmodel, streams=1).to(dev)
    with torch.no_grad():
        pbench(x)
        prof = pmodel._engine.profile(reps=5)
    # the launches outside the engine plans: stem conv, top-k, decode + NMS + OOD gather
    from ood_object_detection_amd.effdet.bench import _post_process
    from ood_object_detection_amd.effdet.anchors import batched_detections
    eng = pmodel._engine
    es = 2 if args.dtype == 'bf16' else 4

    def timed(fn, reps=5):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            r = fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps, r
    with torch.no_grad():
        xc = x.contiguous()
        stem_ms, _ = timed(lambda: eng._stem_call(xc))
        cls_o, box_o = eng.head_views(eng.cls_all, eng.C), eng.head_views(eng.box_all, 4)
        ms_topk, pp = timed(lambda: _post_process(cls_o, box_o, cfg.num_levels, args.classes, cfg.max_detection_points,
                                                  anchor_max=eng.ood_max_logit))
        ct, bt, idx, cl = pp
        ms_det, _ = timed(lambda: batched_detections(ct.reshape(B, -1), bt, bench.anchors.boxes, idx, cl, None, None,
                                                     cfg.max_det_per_image, bool(args.soft_nms)))
    Hs = args.image // 2
    prof.append(('backbone.conv_stem%s' % ('+blocks.0.0.conv_dw' if eng._fuse_stem else ''), eng._stem_meta['kind'],
                 eng._stem_meta['bytes'], eng._stem_meta['flops'], stem_ms))
    # algorithmic bytes of the prefiltered select: the per-anchor maxima + the class rows of ~k anchors (twice)
    prof.append(('_post_process top-k', 'topk', B * eng.N * 4 + 2 * B * cfg.max_detection_points * args.classes * es, 0, ms_topk))
    prof.append(('decode + NMS', 'nms', B * cfg.max_detection_points * 40, 0, ms_det))
    fam = {}
    for what, kind, nbytes, flops, ms in prof:
        f = fam.setdefault(kind, dict(ms=0.0, bytes=0, flops=0, launches=0))
        f['ms'] += ms
        f['bytes'] += nbytes
        f['flops'] += flops
        f['launches'] += 1
    dom = max(fam, key=lambda k: fam[k]['ms'])
    d = fam[dom]
    achieved = d['bytes'] / (d['ms'] * 1e-3) / 1e9
    net_bytes = sum(f['bytes'] for f in fam.values())
    net_ms = sum(f['ms'] for f in fam.values())
    # t_roof = sum over launches of max(flops / P_mfma, bytes / BW_hbm) for the fused launch list
    t_roof_ms = 1e3 * sum(max(flops / MFMA_PEAK_FLOPS, nbytes / (HBM_PEAK_GBS * 1e9)) for _, _, nbytes, flops, _ in prof)
    roofline = {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': None,
                'kernel': dom, 'launches_per_step': d['launches'], 'avg_launch_ms': round(d['ms'] / d['launches'], 4),
                'algorithmic_bytes_per_launch': int(d['bytes'] / d['launches']),
                'network_frac': round(net_bytes / (net_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                'step_frac': round(net_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                't_roof_ms': round(t_roof_ms, 4), 't_roof_over_step': round(t_roof_ms / ms_per_step, 4)}
    # HBM traffic of the dominant family: PMC counters cannot be read from inside this process, so the number
    # comes from the committed rocprofv3 --pmc passes of the same workload (tools/profile_gpu.sh ->
    # tools/pmc_traffic.py); null when the workload is not the profiled one.
    wl = '%s/%d/%d/%s/%d' % (args.model, args.image, B, args.dtype, args.classes)
    tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'pmc_traffic_%s.json' % wl.replace('/', '_'))
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get('kernels_sha16') == kernels_sha16() and tj.get('workload') == wl:
                roofline['traffic'] = int(tj['families'][dom]['bytes'] / d['launches'])
                roofline['traffic_over_algorithmic'] = round(tj['families'][dom]['bytes'] / max(1, d['bytes']), 3)
                roofline['traffic_source'] = 'profiles/%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE passes of this workload on these kernel sources, per launch)' % os.path.basename(tpath)
            else:
                roofline['traffic_stale'] = 'committed PMC file was measured on other kernel sources (%s != %s): re-run tools/profile_gpu.sh' % (tj.get('kernels_sha16'), kernels_sha16())
        except (KeyError, ValueError):
            pass
    if args.profile_out:
        with open(args.profile_out, 'w') as fh:
            fh.write('# per-launch HIP-event times, %s %dx%d batch %d %s\n' % (args.model, args.image, args.image, B, args.dtype))
            fh.write('%-58s %-9s %12s %14s %9s %9s\n' % ('launch', 'kind', 'alg_MB', 'GFLOP', 'ms', 'GB/s'))
            for what, kind, nbytes, flops, ms in prof:
                fh.write('%-58s %-9s %12.2f %14.2f %9.4f %9.1f\n' % (what, kind, nbytes / 1e6, flops / 1e9, ms, nbytes / ms / 1e6))
            fh.write('\n# families\n')
            for k, f in sorted(fam.items(), key=lambda kv: -kv[1]['ms']):
                fh.write('%-10s launches %4d  ms %8.3f  alg_GB %8.3f  GB/s %8.1f  TFLOP/s %7.2f\n' % (
                    k, f['launches'], f['ms'], f['bytes'] / 1e9, f['bytes'] / f['ms'] / 1e6, f['flops'] / f['ms'] / 1e9))
            fh.write('# network (sum of launches) ms %.3f ; timed step ms %.3f (adds top-k/decode/NMS/OOD gather)\n' % (net_ms, ms_per_step))
    # ---- beside the headline (N = 1 only, outside the timed region): the same workload with ONE batch in flight, the float32
    #      (reference-precision) path, and the accuracy of the benched bf16 mode at detection level
    extra = {}

    def side(name, fn):
        """a side measurement must never cost the headline line: record the failure instead"""
        try:
            extra[name] = fn()
        except Exception as e:                                   # noqa: BLE001
            extra[name + '_error'] = '%s: %s' % (type(e).__name__, str(e)[:300])
            torch.cuda.empty_cache()
    if world == 1 and not args.no_extras:
        ks = max(5, min(args.steps, 10))
        if nfl != 1:
            side('in_flight_1_images_per_sec',
                 lambda: round(B * ks / timed_steps(copy.copy(model), x, dev, 1, sub, ks, 2, not args.no_graph, barrier)[0], 1))
        if args.dtype == 'bf16':
            def f32_rate():
                m32 = build_model(args.model, args.image, args.classes).to(dev)
                m32.config.soft_nms = bool(args.soft_nms)
                e32, _, _ = timed_steps(m32, x.float(), dev, nfl, sub, ks, 2, not args.no_graph, barrier)
                return round(B * ks / e32, 1)
            side('f32_images_per_sec', f32_rate)
            torch.cuda.empty_cache()

            def mixed_rate():
                from ood_object_detection_amd.serving import MixedPrecisionEfficientDet
                mm = MixedPrecisionEfficientDet(build_model(args.model, args.image, args.classes).to(dev))
                mm.config.soft_nms = bool(args.soft_nms)
                em, _, _ = timed_steps(mm, x, dev, nfl, sub, ks, 2, not args.no_graph, barrier)
                return round(B * ks / em, 1)
            # the point between the two: bfloat16 backbone, float32 BiFPN + heads (serving.MixedPrecisionEfficientDet)
            side('mixed_images_per_sec', mixed_rate)
            torch.cuda.empty_cache()
            def accurate_rate():
                ma = build_model(args.model, args.image, args.classes).to(dev)
                ma.compute_mode = 'accurate'
                ma.config.soft_nms = bool(args.soft_nms)
                ea, _, _ = timed_steps(ma, x.float(), dev, nfl, sub, ks, 2, not args.no_graph, barrier)
                return round(B * ks / ea, 1)
            # the mode that meets north_star's 1e-3: float32 master weights, two-term bf16 values, 3 MFMAs per multiply
            side('accurate_images_per_sec', accurate_rate)
            torch.cuda.empty_cache()
            xp = torch.randn(4, 3, args.image, args.image, generator=torch.Generator().manual_seed(5))
            side('parity_bf16', lambda: parity_bf16(build_model(args.model, args.image, args.classes), xp, dev, soft_nms=args.soft_nms))
            side('parity_mixed', lambda: parity_bf16(build_model(args.model, args.image, args.classes), xp, dev, soft_nms=args.soft_nms,
                                                      candidate='mixed'))
    if args.auroc_surrogate and world == 1:
        side('auroc_surrogate', lambda: auroc_surrogate(copy.copy(model), args.image, dev, dtype, bs=min(B, 8)))
    cpu = None
    if want_cpu:
        def gpu_check(x1):
            m32 = build_model(args.model, args.image, args.classes)
            m32.load_state_dict(sd_cpu)
            m32 = m32.to(dev)
            co, bo = m32(x1.to(dev))
            return [t.float() for t in co], [t.float() for t in bo], m32.ood_energy.clone()
        try:
            cpu = cpu_baseline(sd_cpu, cfg, args.image, args.classes, gpu_check=gpu_check)
        except Exception as e:                                   # noqa: BLE001
            cpu = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
        if args.dtype == 'bf16' and not args.no_extras and args.model == 'tf_efficientdet_d0':
            # the accuracy of the reduced-precision modes on a WELL-CONDITIONED weight set (the oracle calibrates its BatchNorm
            # statistics, so this belongs to the CPU leg): scores spread over (0, 1), near-ties of top-k / NMS are rare
            def calibrated(candidate):
                mc, xc = calibrated_model()
                return dict(parity_bf16(mc, xc, dev, soft_nms=args.soft_nms, candidate=candidate),
                            weights='BN-calibrated seeded d0 (tests/_models.py), 512 px, class bias -2')
            side('parity_bf16_calibrated', lambda: calibrated('bf16'))
            side('parity_mixed_calibrated', lambda: calibrated('mixed'))
            # the accurate mode against the ORACLE itself (float32 PyTorch on the host), bench weights and calibrated weights
            side('parity_accurate', lambda: parity_accurate_vs_oracle(build_model(args.model, args.image, args.classes), None, dev, args.image,
                                                                         args.classes, soft_nms=args.soft_nms))
            side('parity_accurate_calibrated', lambda: parity_accurate_vs_oracle(*calibrated_model(), dev, 512, 90, soft_nms=args.soft_nms))
    out = {
        'metric': 'images/sec, %s %dpx %s inference + OOD score (DetBenchPredict end-to-end)' % (args.model, args.image, args.dtype),
        'value': round(value, 2), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'bf16x2 (two-term bf16, fp32 accumulate)' if args.dtype == 'accurate' else args.dtype, 'data': 'synthetic',
        'config': {'workload': '%s %dx%d batch=%d/GPU C=%d, DetBenchPredict: backbone+BiFPN+heads+OOD energy/max-logit+top-k(5000)+decode+%s NMS'
                               % (args.model, args.image, args.image, B, args.classes, 'soft' if args.soft_nms else 'hard'),
                   'global_batch': world * B, 'parallelism': 'image-sharded dp%d, no collective' % world,
                   'weights': 'seeded reference init, randomised BN stats, class bias 0', 'launch': launch,
                   'execution': '%d batches in flight (steps alternate between engine instances / streams), each as %d concurrent '
                                'sub-batches' % (nfl, bench.streams or (2 if B >= 16 and B % 2 == 0 else 1)),
                   'detections_per_image_mean': round(float(counts.mean()), 1)},
        'roofline': roofline, 'cpu_baseline': cpu,
    }
    out['config']['images_in_flight'] = nfl * B
    # the limiter of image-sharded inference is the host-side feed, so the rate is quoted with the inputs already in HBM
    # (every rank generates its batch on its own GPU before the timed region); the PCIe-inclusive rate is a separate measurement
    out['config']['inputs'] = ('device-resident (generated on each rank\'s GPU before the timed region); host-to-device copies are not in '
                               'the timed region - PCIe-inclusive rate: tools/pcie_rate.py (profiles/r01_q_pcie_inclusive_rate.json, DESIGN 5)')
    out['ranks'] = {'world_size': world, 'answered_all_reduce': ranks_seen, 'backend': ('rccl' if backend == 'nccl' else backend) if world > 1 else None,
                    'per_rank_images_per_sec': per_rank}
    out.update(extra)
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
