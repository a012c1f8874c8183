"""ORACLE (test infrastructure, never shipped or measured as the product).

CPU fp32 restatement, in plain PyTorch, of the EfficientDet forward pass:
EfficientNet backbone -> BiFPN -> class/box HeadNets.  It is a pure function of a
state-dict (the checkpoint layout of SURVEY §8b) and a config, so it shares no
module code with `ood_object_detection_amd`.

Follows, by reference file:line
  effdet/efficientdet.py:42-83    ConvBnAct2d / SeparableConv2d
  effdet/efficientdet.py:140-177  ResampleFeatureMap (conv before pool, conv_after_downsample=False)
  effdet/efficientdet.py:196-245  FpnCombine ('fastattn' / 'attn' / 'sum')
  effdet/efficientdet.py:261-300  BiFpnLayer (node = combine -> act -> separable conv, no act after)
  effdet/efficientdet.py:303-365  BiFpn (extra levels by 1x1 conv+BN then max-pool)
  effdet/efficientdet.py:368-469  HeadNet (shared convs, per-level BN)
  effdet/efficientdet.py:895-933  EfficientDet.forward modes

PARITY STATUS
  * BiFPN / HeadNet wiring: pinned against the reference's own modules (imported in the build
    container with layer stubs, tools/make_golden.py -> tests/golden/bifpn_head_*.npz).
  * timm EfficientNet backbone and timm's TF-"SAME" conv/pool padding: timm is a third-party
    dependency that is absent from /root/reference (unpinned version; call sites
    effdet/efficientdet.py:17-18,837).  Its published algorithm is restated here from memory:
    PARITY UNPINNED for the backbone.
"""
import math

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------------
# EfficientNet architecture table (timm `efficientnet_b0` arch_def + compound scaling).
# ----------------------------------------------------------------------------------------------
# (type, repeats, kernel, stride, expand, out_channels), se_ratio 0.25 everywhere.
_B0_STAGES = [
    ('ds', 1, 3, 1, 1, 16),
    ('ir', 2, 3, 2, 6, 24),
    ('ir', 2, 5, 2, 6, 40),
    ('ir', 3, 3, 2, 6, 80),
    ('ir', 3, 5, 1, 6, 112),
    ('ir', 4, 5, 2, 6, 192),
    ('ir', 1, 3, 1, 6, 320),
]
# name -> (channel_multiplier, depth_multiplier)
_SCALING = {
    'tf_efficientnet_b0': (1.0, 1.0), 'tf_efficientnet_b1': (1.0, 1.1),
    'tf_efficientnet_b2': (1.1, 1.2), 'tf_efficientnet_b3': (1.2, 1.4),
    'tf_efficientnet_b4': (1.4, 1.8), 'tf_efficientnet_b5': (1.6, 2.2),
}


def _make_divisible(v, divisor=8):
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def backbone_conventions(backbone_name):
    """-> (pad_type, BatchNorm eps) of a timm EfficientNet variant: the `tf_` family uses TF-"SAME" padding and eps 1e-3, the
    PyTorch-trained variants (efficientnet_b0 ...) static symmetric padding ('') and nn.BatchNorm2d's default eps 1e-5 (timm's
    efficientnet.py; absent dependency, restated - see DESIGN.md)"""
    return ('same', 1e-3) if backbone_name.startswith('tf_') else ('', 1e-5)


def efficientnet_spec(backbone_name):
    """Returns (stem_chs, stages) with stages = list of lists of block dicts."""
    cm, dm = _SCALING[backbone_name if backbone_name.startswith('tf_') else 'tf_' + backbone_name]
    stem = _make_divisible(32 * cm)
    stages = []
    in_chs = stem
    for (btype, rep, k, s, e, c) in _B0_STAGES:
        out = _make_divisible(c * cm)
        rep = int(math.ceil(rep * dm))
        blocks = []
        for b in range(rep):
            stride = s if b == 0 else 1
            blocks.append(dict(type=btype, k=k, s=stride, exp=e, cin=in_chs, cout=out,
                               mid=in_chs * e, se=_se_chs(in_chs),
                               residual=(stride == 1 and in_chs == out)))
            in_chs = out
        stages.append(blocks)
    return stem, stages


def _se_chs(in_chs, se_ratio=0.25):
    # timm: make_divisible(in_chs * se_ratio, divisor=1) -> round-half-up to int, min 1
    v = in_chs * se_ratio
    return max(1, int(v + 0.5))


FEATURE_STAGES = (2, 4, 6)   # out_indices=(2,3,4) of timm's feature_info == stages 2,4,6


# ----------------------------------------------------------------------------------------------
# primitive ops
# ----------------------------------------------------------------------------------------------
def same_pad_amounts(size, k, s, d=1):
    total = max((math.ceil(size / s) - 1) * s + (k - 1) * d + 1 - size, 0)
    return total // 2, total - total // 2


def conv2d_pad(x, w, b, stride, pad_type, groups=1):
    k = w.shape[-1]
    if pad_type == 'same':
        if stride == 1 and k % 2 == 1:
            return F.conv2d(x, w, b, stride, (k - 1) // 2, 1, groups)
        pt, pb = same_pad_amounts(x.shape[-2], k, stride)
        pl, pr = same_pad_amounts(x.shape[-1], k, stride)
        x = F.pad(x, [pl, pr, pt, pb])
        return F.conv2d(x, w, b, stride, 0, 1, groups)
    # '' -> symmetric static padding
    return F.conv2d(x, w, b, stride, ((stride - 1) + (k - 1)) // 2, 1, groups)


def maxpool_pad(x, k, s, pad_type):
    if pad_type == 'same':
        pt, pb = same_pad_amounts(x.shape[-2], k, s)
        pl, pr = same_pad_amounts(x.shape[-1], k, s)
        x = F.pad(x, [pl, pr, pt, pb], value=float('-inf'))
        return F.max_pool2d(x, k, s, 0)
    return F.max_pool2d(x, k, s, ((s - 1) + (k - 1)) // 2)


# Test-data helper: when set, every BN sets its running stats to the statistics of the batch it sees
# (momentum 1), which turns seeded random weights into a well-conditioned, "trained-like" network.
CALIBRATE = False


# Training restatement (pretrain.py:168-176, 226-236): BatchNorm layers whose state-dict prefix starts with one of these
# strings run as nn.BatchNorm2d does in training mode - batch statistics, running stats updated in place with
# BN_MOMENTUM (config norm_kwargs momentum .01) - the rest use running statistics.  Default: none (inference).
BN_BATCH_STATS_PREFIXES = ()
BN_MOMENTUM = 0.01


def bn_eval(x, sd, prefix, eps):
    if BN_BATCH_STATS_PREFIXES and prefix.startswith(tuple(BN_BATCH_STATS_PREFIXES)):
        return F.batch_norm(x, sd[prefix + 'running_mean'], sd[prefix + 'running_var'],
                            sd[prefix + 'weight'], sd[prefix + 'bias'], True, BN_MOMENTUM, eps)
    if CALIBRATE:
        F.batch_norm(x, sd[prefix + 'running_mean'], sd[prefix + 'running_var'], None, None, True, 1.0, eps)
    return F.batch_norm(x, sd[prefix + 'running_mean'], sd[prefix + 'running_var'],
                        sd[prefix + 'weight'], sd[prefix + 'bias'], False, 0.0, eps)


def silu(x):
    return x * torch.sigmoid(x)


# ----------------------------------------------------------------------------------------------
# backbone
# ----------------------------------------------------------------------------------------------
def backbone_forward(sd, backbone_name, x, prefix='backbone.', eps=None, pad_type=None, drop_scales=None):
    """drop_scales (training restatement): {flat block index: [B] tensor} = timm's drop_path factor floor(keep + U) / keep of the
    residual branch of that block (timm `drop_path`, layers/drop.py - absent dependency, restated from the published code:
    `x.div(keep_prob) * random_tensor` then `x += shortcut`; per-block rate drop_path_rate * idx / n_blocks)"""
    stem, stages = efficientnet_spec(backbone_name)
    conv_pad, conv_eps = backbone_conventions(backbone_name)
    pad_type = conv_pad if pad_type is None else pad_type           # (the backbone follows ITS name, not the detector's config.pad_type)
    eps = conv_eps if eps is None else eps
    flat_idx = 0
    g = lambda k: sd[prefix + k]
    x = conv2d_pad(x, g('conv_stem.weight'), None, 2, pad_type)
    x = silu(bn_eval(x, sd, prefix + 'bn1.', eps))
    feats = []
    for si, blocks in enumerate(stages):
        for bi, blk in enumerate(blocks):
            p = '%sblocks.%d.%d.' % (prefix, si, bi)
            shortcut = x
            if blk['type'] == 'ds':
                x = conv2d_pad(x, sd[p + 'conv_dw.weight'], None, blk['s'], pad_type, groups=blk['cin'])
                x = silu(bn_eval(x, sd, p + 'bn1.', eps))
                x = _se(x, sd, p + 'se.')
                x = conv2d_pad(x, sd[p + 'conv_pw.weight'], None, 1, pad_type)
                x = bn_eval(x, sd, p + 'bn2.', eps)
            else:
                x = conv2d_pad(x, sd[p + 'conv_pw.weight'], None, 1, pad_type)
                x = silu(bn_eval(x, sd, p + 'bn1.', eps))
                x = conv2d_pad(x, sd[p + 'conv_dw.weight'], None, blk['s'], pad_type, groups=blk['mid'])
                x = silu(bn_eval(x, sd, p + 'bn2.', eps))
                x = _se(x, sd, p + 'se.')
                x = conv2d_pad(x, sd[p + 'conv_pwl.weight'], None, 1, pad_type)
                x = bn_eval(x, sd, p + 'bn3.', eps)
            if blk['residual']:
                if drop_scales is not None and flat_idx in drop_scales:
                    x = x * drop_scales[flat_idx].to(x).view(-1, 1, 1, 1)
                x = x + shortcut
            flat_idx += 1
        if si in FEATURE_STAGES:
            feats.append(x)
    return feats


def _se(x, sd, p):
    s = x.mean((2, 3), keepdim=True)
    s = F.conv2d(s, sd[p + 'conv_reduce.weight'], sd[p + 'conv_reduce.bias'])
    s = silu(s)
    s = F.conv2d(s, sd[p + 'conv_expand.weight'], sd[p + 'conv_expand.bias'])
    return x * torch.sigmoid(s)


def backbone_feature_info(backbone_name):
    _, stages = efficientnet_spec(backbone_name)
    red = 2
    info = []
    for si, blocks in enumerate(stages):
        for blk in blocks:
            red *= blk['s']
        if si in FEATURE_STAGES:
            info.append(dict(num_chs=blocks[-1]['cout'], reduction=red))
    return info


# ----------------------------------------------------------------------------------------------
# BiFPN
# ----------------------------------------------------------------------------------------------
def _resample(x, sd, p, cfg, in_chs, reduction_ratio, eps):
    """ResampleFeatureMap: optional 1x1 conv(+BN) then pool / upsample (efficientdet.py:153-177)."""
    F_ = cfg.fpn_channels
    if in_chs != F_:
        bias = sd.get(p + 'conv.conv.bias')
        x = conv2d_pad(x, sd[p + 'conv.conv.weight'], bias, 1, cfg.pad_type)
        if cfg.apply_resample_bn:
            x = bn_eval(x, sd, p + 'conv.bn.', eps)
    if reduction_ratio > 1:
        s = int(reduction_ratio)
        x = maxpool_pad(x, s + 1, s, cfg.pad_type)
    elif reduction_ratio < 1:
        scale = int(1 // reduction_ratio)
        x = F.interpolate(x, scale_factor=float(scale), mode='nearest')
    return x


def _combine(nodes, edge_weights, method):
    if method == 'fastattn':
        w = F.relu(edge_weights)
        s = w.sum()
        out = torch.stack([(nodes[i] * w[i]) / (s + 0.0001) for i in range(len(nodes))], dim=-1)
    elif method == 'attn':
        out = torch.stack(nodes, dim=-1) * torch.softmax(edge_weights, dim=0)
    elif method == 'sum':
        out = torch.stack(nodes, dim=-1)
    else:
        raise ValueError(method)
    return out.sum(dim=-1)


def _sepconv(x, sd, p, pad_type, bn_eps=None, act=False):
    """SeparableConv2d: dw kxk (no bias) -> pw 1x1 (+bias) -> BN -> act (efficientdet.py:76-83)."""
    x = conv2d_pad(x, sd[p + 'conv_dw.weight'], None, 1, pad_type, groups=x.shape[1])
    x = conv2d_pad(x, sd[p + 'conv_pw.weight'], sd.get(p + 'conv_pw.bias'), 1, pad_type)
    if bn_eps is not None:
        x = bn_eval(x, sd, p + 'bn.', bn_eps)
    return silu(x) if act else x


def bifpn_forward(sd, cfg, feats, fpn_nodes, feature_info, prefix='fpn.'):
    eps = cfg.norm_kwargs['eps']
    x = list(feats)
    info = [dict(f) for f in feature_info]
    # extra coarse levels (BiFpn.__init__ :316-337, forward :362-363)
    for level in range(cfg.num_levels):
        if level < len(info):
            continue
        in_chs = info[-1]['num_chs']
        x.append(_resample(x[-1], sd, '%sresample.%d.' % (prefix, level), cfg, in_chs, 2, eps))
        info.append(dict(num_chs=cfg.fpn_channels, reduction=info[-1]['reduction'] * 2))
    for rep in range(cfg.fpn_cell_repeats):
        for ni, node in enumerate(fpn_nodes):
            p = '%scell.%d.fnode.%d.' % (prefix, rep, ni)
            ins = []
            for off in node['inputs_offsets']:
                if off < len(info):
                    in_chs, in_red = info[off]['num_chs'], info[off]['reduction']
                else:
                    in_chs, in_red = cfg.fpn_channels, fpn_nodes[off - len(info)]['reduction']
                ratio = node['reduction'] / in_red
                ins.append(_resample(x[off], sd, '%scombine.resample.%d.' % (p, off), cfg, in_chs, ratio, eps))
            y = _combine(ins, sd.get(p + 'combine.edge_weights'), node['weight_method'])
            y = silu(y)
            y = _sepconv(y, sd, p + 'after_combine.conv.', cfg.pad_type, bn_eps=eps, act=False)
            x.append(y)
        x = x[-cfg.num_levels:]
        info = [dict(num_chs=cfg.fpn_channels, reduction=n['reduction']) for n in fpn_nodes[-cfg.num_levels:]]
    return x


def head_forward(sd, cfg, feats, prefix):
    eps = cfg.norm_kwargs['eps']
    outs = []
    for level in range(cfg.num_levels):
        x = feats[level]
        for r in range(cfg.box_class_repeats):
            x = _sepconv(x, sd, '%sconv_rep.%d.' % (prefix, r), cfg.pad_type)
            x = bn_eval(x, sd, '%sbn_rep.%d.%d.bn.' % (prefix, r, level), eps)
            x = silu(x)
        outs.append(_sepconv(x, sd, prefix + 'predict.', cfg.pad_type))
    return outs


def efficientdet_forward(sd, cfg, x, fpn_nodes, mode='full_net', drop_scales=None):
    """EfficientDet.forward (efficientdet.py:895-933) for the modes that do not need MetaHead."""
    sd = {k: v.float() for k, v in sd.items() if torch.is_tensor(v)}
    info = backbone_feature_info(cfg.backbone_name)
    if mode == 'bb':
        return backbone_forward(sd, cfg.backbone_name, x, drop_scales=drop_scales)
    if mode in ('full_net', 'fpn', 'supp_bb'):
        feats = backbone_forward(sd, cfg.backbone_name, x, drop_scales=drop_scales)
        activs = bifpn_forward(sd, cfg, feats, fpn_nodes, info)
        if mode == 'fpn':
            return feats, activs
        if mode == 'supp_bb':
            return activs
        return head_forward(sd, cfg, activs, 'class_net.'), head_forward(sd, cfg, activs, 'box_net.')
    if mode == 'only_fpn':
        return bifpn_forward(sd, cfg, x, fpn_nodes, info)
    if mode in ('fpn_and_head', 'not_cls'):
        activs = bifpn_forward(sd, cfg, x, fpn_nodes, info)
        if mode == 'not_cls':
            return activs, head_forward(sd, cfg, activs, 'box_net.')
        return head_forward(sd, cfg, activs, 'class_net.'), head_forward(sd, cfg, activs, 'box_net.')
    if mode == 'head':
        return head_forward(sd, cfg, x, 'class_net.'), head_forward(sd, cfg, x, 'box_net.')
    raise ValueError('oracle: unsupported mode %r' % (mode,))


def ood_scores(cls_outs, num_classes):
    """Per-anchor OOD scores (build-defined, no reference symbol):
    energy = -logsumexp_c z, max_logit = max_c z over the class logits of each anchor.
    Input: list of [B, A*C, H, W]; output two [B, N] tensors in the anchor order of
    effdet/bench.py:36-38."""
    B = cls_outs[0].shape[0]
    z = torch.cat([o.permute(0, 2, 3, 1).reshape(B, -1, num_classes) for o in cls_outs], 1).float()
    return -torch.logsumexp(z, dim=2), z.amax(dim=2)


def meta_head_forward(conv_dw_rep, conv_pw_rep, conv_pb_rep, bn_rep_w, bn_rep_b, predict, x, level_offset=0,
                      predict_class=None, eps=1e-5):
    """MetaHead.forward (effdet/efficientdet.py:636-695) with explicit weight lists in the reference's order
    (bn lists are level-major: index level * num_layers + rep).  Returns (outputs, x_pred activations[, class_outputs]).
    Pinned: tools/make_golden.py instantiates the reference class on the CPU (its `.to('cuda')` calls neutralised in that
    script only) and tests/golden/meta_nets.npz holds its outputs; tests/test_oracle_golden.py checks this restatement
    against them."""
    num_layers = len(conv_dw_rep)
    outputs, activs, class_outputs = [], [], []
    for level in range(level_offset, len(x)):
        x_level = x[level].float()
        bn_w_lev = bn_rep_w[level * num_layers:(level + 1) * num_layers]
        bn_b_lev = bn_rep_b[level * num_layers:(level + 1) * num_layers]
        for conv_dw, conv_pw, conv_pb, bn_w, bn_b in zip(conv_dw_rep, conv_pw_rep, conv_pb_rep, bn_w_lev, bn_b_lev):
            x_level = F.pad(x_level, (1, 1, 1, 1))
            x_level = F.conv2d(x_level, conv_dw.float(), groups=conv_dw.shape[0], padding=(0, 0))
            x_level = F.conv2d(x_level, conv_pw.float(), bias=conv_pb.float())
            x_level = F.batch_norm(x_level, None, None, bn_w.float(), bn_b.float(), training=True, eps=eps)
            x_level = x_level * torch.sigmoid(x_level)
        x_pred = F.pad(x_level, (1, 1, 1, 1))
        x_pred = F.conv2d(x_pred, predict[0].float(), groups=predict[0].shape[0])
        activs.append(x_pred)
        outputs.append(F.conv2d(x_pred, predict[1].float(), bias=predict[2].float()))
        if predict_class is not None:
            class_outputs.append(F.conv2d(x_pred, predict_class[0].float(), bias=predict_class[1].float()))
    return (outputs, activs, class_outputs) if predict_class is not None else (outputs, activs)


def anchor_net_forward(sd, x, num_levels, eps=1e-3, prefix=''):
    """AnchorNet.forward (effdet/efficientdet.py:817-830) from a state dict: per level conv_rep[i] (SeparableConv, TF-SAME
    3x3 = symmetric pad 1) -> bn_rep[i][level] (eval) -> Swish, then anchor_out.  Pinned by tests/golden/meta_nets.npz (reference
    AnchorNet run by tools/make_golden.py with the absl stub supplying its FLAGS)."""
    outs = []
    n_rep = len({k.split('.')[1] for k in sd if k.startswith(prefix + 'conv_rep.')})
    for level in range(len(x)):
        t = x[level].float()
        for i in range(n_rep):
            p = '%sconv_rep.%d.' % (prefix, i)
            t = F.conv2d(F.pad(t, (1, 1, 1, 1)), sd[p + 'conv_dw.weight'].float(), groups=t.shape[1])
            t = F.conv2d(t, sd[p + 'conv_pw.weight'].float(), bias=sd[p + 'conv_pw.bias'].float() if (p + 'conv_pw.bias') in sd else None)
            b = '%sbn_rep.%d.%d.bn.' % (prefix, i, level)
            t = F.batch_norm(t, sd[b + 'running_mean'].float(), sd[b + 'running_var'].float(), sd[b + 'weight'].float(),
                             sd[b + 'bias'].float(), training=False, eps=eps)
            t = t * torch.sigmoid(t)
        p = prefix + 'anchor_out.'
        t = F.conv2d(F.pad(t, (1, 1, 1, 1)), sd[p + 'conv_dw.weight'].float(), groups=t.shape[1])
        outs.append(F.conv2d(t, sd[p + 'conv_pw.weight'].float(), bias=sd[p + 'conv_pw.bias'].float()))
    return outs


def projection_forward(weights, x):
    """ProjectionNet.forward (efficientdet.py:762): Linear(bias=False) + ReLU chain, last layer without ReLU."""
    t = x.float()
    for i, w in enumerate(weights):
        t = F.linear(t, w.float())
        if i + 1 < len(weights):
            t = F.relu(t)
    return t


def weighted_median(embds, confs):
    """ProjectionNet.weighted_median (efficientdet.py:748-760), the reference's torch ops verbatim in meaning."""
    conf_sum = confs.sum()
    sorted_elems, sorted_idxs = torch.sort(embds, dim=0, stable=True)
    sorted_confs = confs[sorted_idxs.transpose(0, 1)].transpose(0, 1)
    cum_sum = torch.cumsum(sorted_confs, dim=0)
    mask = (cum_sum >= conf_sum / 2).long()
    median_idxs = torch.argmax(mask, dim=0).view(1, -1)
    return torch.gather(sorted_elems, 0, median_idxs), conf_sum
