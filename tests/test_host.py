"""Host-side logic (no GPU): config tables, anchors, state-dict layout, C ABI exports."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _norm(v):
    if isinstance(v, (list, tuple)):
        return [_norm(x) for x in v]
    if isinstance(v, dict):
        return {k: _norm(x) for k, x in v.items()}
    return v


@pytest.mark.parametrize('name', ['tf_efficientdet_d0', 'tf_efficientdet_d1', 'tf_efficientdet_d2', 'tf_efficientdet_d3',
                                  'tf_efficientdet_d4', 'tf_efficientdet_d5'])
def test_config_matches_reference(golden, name):
    from ood_object_detection_amd.effdet.config import get_efficientdet_config
    ref = json.loads(str(golden('config')[name]))
    h = get_efficientdet_config(name)
    assert sorted(h.keys()) == sorted(ref.keys())
    for k in ref:
        assert _norm(h[k]) == _norm(ref[k]), k


def test_unknown_model_raises_keyerror():
    from ood_object_detection_amd.effdet.config import get_efficientdet_config
    with pytest.raises(KeyError):
        get_efficientdet_config('nope')


def test_bifpn_graph_matches_reference(golden):
    from ood_object_detection_amd.effdet.config import get_fpn_config
    g = golden('config')
    for key, (name, lo, hi) in {'bifpn_fa_3_7': ('bifpn_fa', 3, 7), 'bifpn_sum_3_8': ('bifpn_sum', 3, 8)}.items():
        ref = json.loads(str(g[key]))
        got = [dict(n) for n in get_fpn_config(name, lo, hi).nodes]
        assert _norm(got) == _norm(ref)


@pytest.mark.parametrize('size', [128, 512, 640, 768, 1024])
def test_product_anchors_bit_exact(golden, size):
    import hashlib
    from ood_object_detection_amd.effdet.anchors import Anchors
    g = golden('anchors')
    b = Anchors(3, 7, 3, [(1.0, 1.0), (1.4, 0.7), (0.7, 1.4)], 4.0, (size, size)).boxes
    assert hashlib.sha256(b.numpy().tobytes()).digest() == g['sha256_%d' % size].tobytes()


def test_product_anchors_odd(golden):
    from ood_object_detection_amd.effdet.anchors import Anchors
    b = Anchors(3, 6, 2, [1.0, 2.0, 0.5], [4.0, 3.0, 4.0, 5.0], (128, 256)).boxes
    assert np.array_equal(b.numpy(), golden('anchors')['odd_full'])
    with pytest.raises(AssertionError):
        Anchors(3, 7, 3, [(1.0, 1.0)], 4.0, (100, 128))


@pytest.mark.parametrize('tag,name', [('d0', 'tf_efficientdet_d0'), ('d1', 'tf_efficientdet_d1')])
def test_state_dict_layout_matches_reference(golden, tag, name):
    """fpn.* / class_net.* / box_net.* keys and shapes equal the reference module tree's."""
    from ood_object_detection_amd.effdet.config import get_efficientdet_config
    from ood_object_detection_amd.effdet.efficientdet import EfficientDet
    g = golden('bifpn_head')
    size, ncls, _ = [int(v) for v in g[tag + '_meta']]
    cfg = get_efficientdet_config(name)
    cfg.image_size = (size, size)
    cfg.num_classes = ncls
    sd = EfficientDet(cfg, pretrained_backbone=False).state_dict()
    mine = {k: list(v.shape) for k, v in sd.items() if not k.endswith('num_batches_tracked')}
    ref = {str(k): json.loads(str(s)) for k, s in zip(g[tag + '_keys'], g[tag + '_shapes'])}
    assert mine == ref
    assert list(mine.keys()) == list(ref.keys())       # same order too


@pytest.mark.parametrize('tag,name', [('d0', 'efficientdet_d0'), ('d1', 'efficientdet_d1')])
def test_state_dict_layout_matches_reference_pad0(golden, tag, name):
    """the PyTorch-trained family (pad_type '', redundant_bias False: no conv_pw / resample-conv biases in front of a BN): keys,
    shapes and order equal the reference module tree's, so rwightman's efficientdet_d0 / d1 checkpoints load strictly"""
    from ood_object_detection_amd.effdet.config import get_efficientdet_config
    from ood_object_detection_amd.effdet.efficientdet import EfficientDet
    g = golden('bifpn_head_pad0')
    size, ncls, _ = [int(v) for v in g[tag + '_meta']]
    cfg = get_efficientdet_config(name)
    cfg.image_size = (size, size)
    cfg.num_classes = ncls
    m = EfficientDet(cfg, pretrained_backbone=False)
    sd = m.state_dict()
    mine = {k: list(v.shape) for k, v in sd.items() if not k.endswith('num_batches_tracked')}
    ref = {str(k): json.loads(str(s)) for k, s in zip(g[tag + '_keys'], g[tag + '_shapes'])}
    assert mine == ref
    assert list(mine.keys()) == list(ref.keys())
    assert m.backbone.pad_type == '' and m.backbone.bn1.eps == 1e-5 and m.backbone.bn1.momentum == 0.1      # timm's non-tf defaults
    assert m.fpn.cell[0].fnode[0].after_combine.conv.bn.eps == 1e-3                                       # config.norm_kwargs


def test_scripts_default_models_construct():
    """pretrain.py:81-112 and infer.py:119-149 build `default_detection_model_configs()` updated with these dicts and call
    EfficientDet(h): `--model d0` (the default) and `--model d1` are efficientdet_d0 / d1 on efficientnet_b0 / b1 with
    pad_type '' and redundant_bias False; `--model d3` is tf_efficientdet_d3."""
    import torch
    from ood_object_detection_amd.effdet.config import default_detection_model_configs
    from ood_object_detection_amd.effdet.efficientdet import EfficientDet
    dicts = [dict(name='efficientdet_d0', backbone_name='efficientnet_b0', image_size=(640, 640), fpn_channels=64, fpn_cell_repeats=3,
                  box_class_repeats=3, pad_type='', redundant_bias=False, backbone_args=dict(drop_path_rate=0.2)),
             dict(name='efficientdet_d1', backbone_name='efficientnet_b1', image_size=(640, 640), fpn_channels=88, fpn_cell_repeats=4,
                  box_class_repeats=3, pad_type='', redundant_bias=False, backbone_args=dict(drop_path_rate=0.2)),
             dict(name='tf_efficientdet_d3', backbone_name='tf_efficientnet_b3', image_size=(640, 640), fpn_channels=160,
                  fpn_cell_repeats=6, box_class_repeats=4, backbone_args=dict(drop_path_rate=0.2))]
    for d in dicts:
        h = default_detection_model_configs()
        h.update(d)
        h.num_levels = h.max_level - h.min_level + 1
        m = EfficientDet(h)
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        m.load_state_dict(sd, strict=True)
        has_bias = any(k.endswith('conv_rep.0.conv_pw.bias') for k in sd)
        assert has_bias == (d.get('redundant_bias', True))
        assert m.backbone.pad_type == ('same' if d['backbone_name'].startswith('tf_') else '')


def test_reset_head_and_param_count():
    from ood_object_detection_amd.effdet.factory import create_model
    m = create_model('tf_efficientdet_d0', num_classes=90)
    assert m.class_net.predict.conv_pw.weight.shape == (810, 64, 1, 1)
    assert abs(float(m.class_net.predict.conv_pw.bias[0]) + np.log(99.0)) < 1e-6
    n = sum(p.numel() for p in m.parameters())
    assert n == 3880067                                 # the published effdet table for tf_efficientdet_d0 (90 classes)


def test_weights_token_sees_replaced_parameters_and_in_place_updates():
    """The engine's packed weights are rebuilt when `weights_token()` changes: in-place updates (version counters), a swapped
    sub-module, and a REPLACED Parameter object (`m.weight = nn.Parameter(...)`: the cached tensor list must not keep the old one)."""
    import torch
    from ood_object_detection_amd.effdet.factory import create_model
    m = create_model('tf_efficientdet_d0', num_classes=20)
    t0 = m.weights_token()
    assert m.weights_token() == t0
    conv = m.fpn.cell[1].fnode[3].after_combine.conv.conv_pw
    with torch.no_grad():
        conv.weight.add_(1.0)
    t1 = m.weights_token()
    assert t1 != t0
    conv.weight = torch.nn.Parameter(torch.zeros_like(conv.weight))          # a new object, version counter 0 again
    t2 = m.weights_token()
    assert t2 != t1
    with torch.no_grad():
        conv.weight.mul_(2.0)                                                # the NEW parameter is the one being watched
    assert m.weights_token() != t2


@pytest.mark.parametrize('name,count,feat_chs', [('tf_efficientdet_d0', 3880067, [40, 112, 320]), ('tf_efficientdet_d1', 6625898, [40, 112, 320]),
                                                 ('tf_efficientdet_d2', 8097039, [48, 120, 352]), ('tf_efficientdet_d3', 12032296, [48, 136, 384]),
                                                 ('tf_efficientdet_d4', 20723675, [56, 160, 448])])
def test_param_counts_equal_the_published_table(name, count, feat_chs):
    """architecture pin for the (absent) timm backbone: the exact parameter counts rwightman/efficientdet-pytorch publishes
    for 90 classes, and the channels of the three backbone feature maps the BiFPN consumes"""
    from ood_object_detection_amd.effdet.factory import create_model
    m = create_model(name, num_classes=90)
    assert sum(p.numel() for p in m.parameters()) == count
    assert [f['num_chs'] for f in m.fpn.in_feature_info[:3]] == feat_chs
    assert [f['reduction'] for f in m.fpn.in_feature_info[:3]] == [8, 16, 32]


def test_meta_nets_match_reference_layout(golden):
    """MetaHead parameter names / order, ProjectionNet encoding tables + layer shapes, AnchorNet state-dict layout against the
    reference classes (fixture meta_nets.npz), and the scripts' default supp_level_offset = 2 (infer.py:94, pretrain.py:63)"""
    from _seeded import meta_nets_case
    from ood_object_detection_amd.effdet.config import get_efficientdet_config
    from ood_object_detection_amd.effdet.efficientdet import AnchorNet, EfficientDet, MetaHead, ProjectionNet
    g = golden('meta_nets')
    c = meta_nets_case(g)
    cfg = get_efficientdet_config('tf_efficientdet_d0')
    mh = MetaHead(cfg, pretrain_init=c['init'])
    assert [n for n, _ in mh.named_parameters()] == [str(n) for n in g['mh_param_names']]
    fw = mh.conv_dw_rep + mh.conv_pw_rep + mh.conv_pb_rep + mh.predict + mh.bn_rep_w + mh.bn_rep_b
    assert len(fw) == int(g['mh_n_fast'])
    for depth in (2, 3, 4):
        pn = ProjectionNet(cfg, 128, proj_depth=depth)
        lin = [m for m in pn.projection if isinstance(m, torch.nn.Linear)]
        assert [m.weight.shape[0] for m in lin] + [lin[0].weight.shape[1]] == [int(v) for v in g['pn%d_dims' % depth]]
    for k in ('anch_enc', 'cell_enc', 'lev_enc'):
        assert np.array_equal(getattr(pn, k).numpy(), g['pn_' + k])
    pn = ProjectionNet(cfg, 128, dot_mult=3.0, dot_add=3.0)
    assert [float(pn.dot_mult), float(pn.dot_add)] == [float(v) for v in g['pn_dot']]
    for layers in (3, 1):
        sd = AnchorNet(cfg, num_anch_layers=layers).state_dict()
        mine = {k: list(v.shape) for k, v in sd.items() if not k.endswith('num_batches_tracked')}
        ref = {str(k): json.loads(str(s)) for k, s in zip(g['an%d_keys' % layers], g['an%d_shapes' % layers])}
        assert mine == ref and list(mine.keys()) == list(ref.keys())
    cfg.num_classes = 3
    assert EfficientDet(cfg, pretrained_backbone=False).supp_level_offset == int(g['supp_level_offset_default']) == 2


def test_stochastic_depth_rates_and_oracle_semantics():
    """timm's stochastic depth as the configs ask for it (backbone_args drop_path_rate = 0.2, pretrain.py:49,94): per-block rate
    = rate * block index / block count on blocks with a residual, and the oracle's drop factor is timm's literal drop_path
    expression `x.div(keep) * floor(keep + U)`"""
    from ood_object_detection_amd.effdet.factory import create_model
    from oracle import model as om
    m = create_model('tf_efficientdet_d0', num_classes=3)
    assert m.backbone.drop_path_rate == 0.2
    rates = m.backbone.block_drop_rates()
    stem, stages = m.backbone.arch
    flat = [b for blocks in stages for b in blocks]
    assert len(rates) == len(flat) == 16
    for i, (r, b) in enumerate(zip(rates, flat)):
        assert r == (0.2 * i / 16 if b['residual'] else 0.0)
    # oracle: x * scale + shortcut with scale = floor(keep + U) / keep  ==  timm's  x.div(keep) * floor(keep + U) + shortcut
    torch.manual_seed(0)
    keep = 1.0 - rates[2]
    u = torch.rand(4)
    rt = torch.floor(keep + u)
    xb = torch.randn(4, 5, 3, 3)
    assert torch.allclose(xb.div(keep) * rt.view(4, 1, 1, 1), xb * (rt / keep).view(4, 1, 1, 1), rtol=1e-6, atol=0)


def test_backbone_arch_tables():
    from ood_object_detection_amd.backbone import efficientnet_arch
    feats = {}
    for name in ('tf_efficientnet_b0', 'tf_efficientnet_b2', 'tf_efficientnet_b4'):
        stem, stages = efficientnet_arch(name)
        feats[name] = [stages[i][-1]['cout'] for i in (2, 4, 6)]
    # SURVEY §8 a3
    assert feats == {'tf_efficientnet_b0': [40, 112, 320], 'tf_efficientnet_b2': [48, 120, 352],
                     'tf_efficientnet_b4': [56, 160, 448]}


def test_product_has_no_cpu_fallback():
    """CPU tensors must fail loudly, never silently compute."""
    from ood_object_detection_amd.effdet.factory import create_model
    from ood_object_detection_amd.effdet.bench import _post_process
    m = create_model('tf_efficientdet_d0', num_classes=3, image_size=(128, 128))
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 128, 128))
    with pytest.raises(RuntimeError):
        _post_process([torch.zeros(1, 27, 4, 4)], [torch.zeros(1, 36, 4, 4)], 1, 3, 10)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, 'ood_object_detection_amd')
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, re.M), f


def test_c_abi_exports_every_declared_symbol():
    """libeffdet_hip.so loads on a GPU-less host and exports exactly what include/effdet_hip.h declares."""
    from ood_object_detection_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    header = open(os.path.join(ROOT, 'include', 'effdet_hip.h')).read()
    declared = set(re.findall(r'\b(effdet_[a-z0-9_]+)\s*\(', header))
    assert declared == set(_lib.SIGNATURES.keys())
    import torch  # noqa: F401  (share torch's HIP runtime, see _lib.load)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.effdet_abi_version() == 1
    # pure-host helpers are callable without a GPU
    assert lib.effdet_dwconv_blocks_per_image(80, 80, 240) > 0
    lib.effdet_topk_workspace_bytes.restype = ctypes.c_longlong
    lib.effdet_topk_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_longlong]
    assert lib.effdet_topk_workspace_bytes(4, 1000) > 4 * (16384 * 8 + 1000 * 8)


@pytest.mark.parametrize('sizes', [(640, 128), (500, 128), (37, 48), (48, 37), (1280, 128), (128, 128)])
def test_pil_coefficient_tables_match_oracle(sizes):
    """The product's vectorised Pillow coefficient tables equal the oracle's scalar restatement (pinned against PIL)."""
    import numpy as np
    from oracle import preprocess as opre
    from ood_object_detection_amd.effdet.preprocess import _pil_bilinear_tables
    a, b = sizes
    if a == b:
        return
    ob, ok = opre.pil_bilinear_coeffs(a, b)
    pb, pk = _pil_bilinear_tables(a, b)
    assert np.array_equal(ob, pb) and np.array_equal(ok, pk)


def test_training_dispatch_and_no_cpu_fallback():
    """EfficientDet.forward picks the differentiable path like the reference does implicitly (module in training mode + grad
    mode + trainable parameters, pretrain.py:226-236) and that path, too, refuses to run on the CPU"""
    import torch
    from ood_object_detection_amd.effdet.factory import create_model
    m = create_model('tf_efficientdet_d0', num_classes=5, image_size=(128, 128))
    assert m.training and m.wants_autograd()
    with torch.no_grad():
        assert not m.wants_autograd()
    m.eval()
    assert not m.wants_autograd()
    m.autograd = True
    assert m.wants_autograd()
    m.autograd = None
    m.train()
    for p in m.parameters():
        p.requires_grad_(False)
    assert not m.wants_autograd()
    for p in m.parameters():
        p.requires_grad_(True)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 128, 128), mode='bb')            # CPU tensors / CPU model: no fallback


def test_pretrain_step_needs_gpu():
    import torch
    from ood_object_detection_amd.effdet.factory import create_model
    from ood_object_detection_amd.pretrain import PretrainStep
    m = create_model('tf_efficientdet_d0', num_classes=5, image_size=(128, 128))
    with pytest.raises(RuntimeError):
        PretrainStep(m)                                        # FlatAdam: float32 parameters on one GPU


def test_script_import_lines_resolve():
    """every `from effdet... import ...` line of infer.py (:11-18) and pretrain.py (:9-16) resolves against this package"""
    import importlib
    wanted = {'effdet.efficientdet': ['EfficientDet', 'AnchorNet', 'ProjectionNet', 'MetaHead'], 'effdet.helpers': ['load_pretrained'],
              'effdet.config.model_config': ['default_detection_model_configs'], 'effdet.distributed': ['all_gather_container'],
              'effdet.bench': ['_post_process'], 'effdet.anchors': ['Anchors', 'AnchorLabeler', 'generate_detections'],
              'effdet.loss': ['DetectionLoss', 'SupportLoss', 'smooth_l1_loss', 'l2_loss', 'cosine_loss'],
              'effdet.evaluation.detection_evaluator': ['ObjectDetectionEvaluator']}
    for mod, names in wanted.items():
        m = importlib.import_module(mod.replace('effdet', 'ood_object_detection_amd.effdet', 1))
        for n in names:
            assert hasattr(m, n), (mod, n)


def test_effdet_importable_under_the_reference_name():
    """with `<repo>/ood_object_detection_amd` first on sys.path the reference's import names resolve to
    this package - one set of module objects under both spellings"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import effdet.factory, effdet.loss, effdet.bench, effdet.anchors, effdet.soft_nms, effdet.config\n"
            "from effdet.evaluation.detection_evaluator import ObjectDetectionEvaluator\n"
            "from effdet.efficientdet import EfficientDet\n"
            "import ood_object_detection_amd.effdet.efficientdet as real\n"
            "assert EfficientDet is real.EfficientDet and effdet.bench is sys.modules['ood_object_detection_amd.effdet.bench']\n"
            "print('ALIAS_OK')\n") % os.path.join(root, 'ood_object_detection_amd')
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd='/tmp', timeout=300)
    assert r.returncode == 0 and 'ALIAS_OK' in r.stdout, r.stderr[-2000:]


def test_create_model_checkpoint_round_trip(tmp_path):
    """factory.create_model(..., checkpoint_path=, checkpoint_ema=) (effdet/factory.py:39-54, timm load_checkpoint): bare state
    dicts, {'state_dict': ...} / {'state_dict_ema': ...} wrappers, DataParallel 'module.' prefixes; a different class count
    resets only class_net.predict.conv_pw (efficientdet.py:854-886); strict loading rejects a wrong layout"""
    import torch
    from ood_object_detection_amd.effdet.factory import create_model
    torch.manual_seed(3)
    src = create_model('tf_efficientdet_d0', num_classes=7, image_size=(128, 128))
    sd = {k: v.clone() for k, v in src.state_dict().items()}
    p1, p2, p3 = str(tmp_path / 'bare.pth'), str(tmp_path / 'wrapped.pth'), str(tmp_path / 'ema.pth')
    torch.save(sd, p1)
    torch.save({'state_dict': {'module.' + k: v for k, v in sd.items()}, 'epoch': 3}, p2)
    torch.save({'state_dict': {k: torch.zeros_like(v) for k, v in sd.items()}, 'state_dict_ema': sd}, p3)
    for path, ema in ((p1, False), (p2, False), (p3, True)):
        bench = create_model('tf_efficientdet_d0', bench_task='predict', num_classes=7, checkpoint_path=path, checkpoint_ema=ema,
                             image_size=(128, 128))
        got = bench.model.state_dict()
        assert set(got) == set(sd)
        assert all(torch.equal(got[k], sd[k]) for k in sd)
    m_def = create_model('tf_efficientdet_d0', image_size=(128, 128))     # the fork's default: FLAGS.pretrain_classes = 400 (model_config.py:30)
    assert m_def.class_net.predict.conv_pw.weight.shape[0] == 9 * 400
    with pytest.raises(RuntimeError):
        create_model('tf_efficientdet_d0', num_classes=5, checkpoint_path=p1, image_size=(128, 128))   # 7-class checkpoint, 5-class head
    with pytest.raises(KeyError):
        create_model('tf_efficientdet_dx')


def test_roofline_table_matches_survey_figures():
    """tools/roofline_table.py: d0 / 640 / C=90 is 3.89 GMAC = 7.79 GFLOP per image (the survey's layer table and
    the paper's 2.5 B at 512 px), N = 76 725 anchors; the fused launch list moves ~155 MB per image"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('roofline_table', os.path.join(root, 'tools', 'roofline_table.py'))
    rt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rt)
    L, geo = rt.layers('tf_efficientdet_d0', 640, 90, 2)
    macs = sum(l[1] for l in L)
    assert geo['N'] == 76725 and geo['P'] == 8525
    assert abs(macs / 1e9 - 3.894) < 0.005
    L512, _ = rt.layers('tf_efficientdet_d0', 512, 90, 2)
    assert abs(sum(l[1] for l in L512) / 1e9 - 2.49) < 0.01
    groups = {}
    for name, m, nb, g, ob in L:
        groups.setdefault(g, [0, []])
        groups[g][0] += nb
        groups[g][1].append(ob)
    fused = sum(nb - 2 * sum(obs[:-1]) for nb, obs in groups.values())
    assert 140e6 < fused < 170e6


def test_only_checkers_import_the_oracle():
    """oracle/ is test infrastructure: nothing in the product package or in tools/ may import it (bench.py may, in its
    cpu_baseline leg only)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pat = re.compile(r'^\s*(from\s+oracle\b|import\s+oracle\b)', re.M)
    offenders = []
    for sub in ('ood_object_detection_amd', 'tools'):
        for dirpath, _, files in os.walk(os.path.join(root, sub)):
            for f in files:
                if f.endswith('.py') and pat.search(open(os.path.join(dirpath, f)).read()):
                    offenders.append(os.path.join(dirpath, f))
    assert not offenders, offenders
    bench_src = open(os.path.join(root, 'bench.py')).read()
    body = bench_src[bench_src.index('def cpu_baseline'):bench_src.index('def main')]
    assert len(pat.findall(bench_src)) == len(pat.findall(body)) > 0        # every oracle import of bench.py sits in cpu_baseline


def test_episode_losses_match_reference(golden):
    """cosine_loss / smooth_l1_loss / l2_loss / SupportLoss (imported by infer.py:17, pretrain.py:15) against the reference's own
    functions (fixture aux_losses.npz); plain tensor expressions, so they run on the CPU here"""
    from _seeded import seeded_array
    from ood_object_detection_amd.effdet.config import get_efficientdet_config
    from ood_object_detection_amd.effdet.loss import SupportLoss, cosine_loss, l2_loss, smooth_l1_loss
    g = golden('aux_losses')
    x = torch.from_numpy(seeded_array(71, 'x', (400,), scale=0.8))
    t = torch.from_numpy((seeded_array(71, 't', (400,)) > 0.3).astype(np.float32) * 2 - 1)
    w = torch.from_numpy(seeded_array(71, 'w', (400,), kind='uniform'))
    tgt = torch.from_numpy(seeded_array(71, 'tgt', (400,), scale=0.5))
    close = lambda a, b: np.allclose(np.asarray(a.detach()), b, rtol=1e-6, atol=1e-7)
    assert close(cosine_loss(x, t, margin=0.), g['cos_0']) and close(cosine_loss(x, t, margin=0.2), g['cos_m'])
    assert close(torch.stack(smooth_l1_loss(x, tgt, beta=1. / 9, weights=w)), g['sl1_b9'])
    assert close(smooth_l1_loss(x, tgt, beta=1. / 9, weights=w, size_average=True), g['sl1_b9_mean'])
    assert close(smooth_l1_loss(x, tgt, beta=0.0, size_average=True), g['sl1_b0_mean'])
    with pytest.raises(UnboundLocalError):             # the reference's pure-L1 branch never defines `err` (loss.py:132-147)
        smooth_l1_loss(x, tgt, beta=0.0, weights=w)
    assert close(torch.stack(l2_loss(x, tgt, weights=w)), g['l2'])
    cfg = get_efficientdet_config('tf_efficientdet_d0')
    cfg.num_classes = 1
    sizes = [int(v) for v in g['meta']]
    co = [torch.from_numpy(seeded_array(72, 'co%d' % i, (2, 9, s, s), scale=1.5)) for i, s in enumerate(sizes)]
    ct = [torch.from_numpy(seeded_array(72, 'ct%d' % i, (2, 9, s, s), kind='uniform')) for i, s in enumerate(sizes)]
    npos = torch.tensor([3., 5.])
    for lt in ('ce', 'mse'):
        for tag, alpha, ls in (('a25', 0.25, 0.0), ('none', None, 0.0), ('ls', 0.25, 0.1)):
            cfg.label_smoothing = ls
            leaf = [c.clone().requires_grad_() for c in co]
            v = SupportLoss(cfg, lt)(leaf, ct, npos, alpha)
            assert close(v, g['sup_%s_%s' % (lt, tag)])
            assert close(torch.autograd.grad(v, leaf)[0], g['sup_%s_%s_g0' % (lt, tag)])
