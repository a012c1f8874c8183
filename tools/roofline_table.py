#!/usr/bin/env python
"""Per-layer algorithmic work of the EfficientDet forward: MAC_l and compulsory bytes_l per image for every
layer of tf_efficientdet_d0 / d2 / d4 (or any model in the config table) at a given image size, class count and element
size, plus the two roofline bounds the survey asks for:

  t_roof(per-layer) = sum_l max(2*MAC_l / P_mfma, bytes_l / BW_hbm),  bytes_l = (in_l + out_l)*es + weights_l*es, BN / act
                      folded, SE counted as one extra read + write of the expanded tensor              (the survey's formula)
  t_roof(fused)     = the same over the fused launch list this build runs: stem+dw0, MBConv front half (expand -> dw with the
                      expanded tensor in LDS), project GEMM (+SE gate, +residual), BiFPN node / head layer as one kernel

CPU only, no GPU and no weights needed:  python tools/roofline_table.py [--model tf_efficientdet_d0 --image 640 --classes 90]
"""
import argparse
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

HBM = 8.0e12          # B/s   (MI355X_MICROARCH.md)
MFMA = 2.5e15         # FLOP/s dense bf16


def _out(n, s):
    return (n + s - 1) // s


def layers(model, image, classes, es):
    """-> list of (name, macs, bytes_unfused, fused_group, output_bytes); per image."""
    from ood_object_detection_amd.backbone import efficientnet_arch
    from ood_object_detection_amd.effdet.config import get_efficientdet_config, get_fpn_config
    cfg = get_efficientdet_config(model)
    stem, stages = efficientnet_arch(cfg.backbone_name)
    F, A = cfg.fpn_channels, len(cfg.aspect_ratios) * cfg.num_scales
    L = []
    h = w = _out(image, 2)
    L.append(('backbone.conv_stem', 27 * h * w * stem, (3 * image * image + h * w * stem) * es + 27 * stem * es, 'stem', h * w * stem * es))
    feats = []
    for si, blocks in enumerate(stages):
        for bi, b in enumerate(blocks):
            p = 'backbone.blocks.%d.%d.' % (si, bi)
            ho, wo = _out(h, b['s']), _out(w, b['s'])
            g = 'front' + p if not (si == 0 and bi == 0) else 'stem'
            if b['type'] == 'ir':
                L.append((p + 'conv_pw', h * w * b['cin'] * b['mid'], (h * w * (b['cin'] + b['mid']) + b['cin'] * b['mid']) * es, g, h * w * b['mid'] * es))
            L.append((p + 'conv_dw', b['k'] ** 2 * ho * wo * b['mid'], (h * w + ho * wo) * b['mid'] * es + b['k'] ** 2 * b['mid'] * 4, g, ho * wo * b['mid'] * es))
            L.append((p + 'se', 2 * b['mid'] * b['se'], 2 * ho * wo * b['mid'] * es + 2 * b['mid'] * b['se'] * 4, 'proj' + p, ho * wo * b['mid'] * es))
            L.append((p + ('conv_pwl' if b['type'] == 'ir' else 'conv_pw'), ho * wo * b['mid'] * b['cout'],
                      (ho * wo * (b['mid'] + b['cout'] * (2 if b['residual'] else 1)) + b['mid'] * b['cout']) * es, 'proj' + p, ho * wo * b['cout'] * es))
            h, w = ho, wo
        if si in (2, 4, 6):
            feats.append((h, w, blocks[-1]['cout']))
    hw = [(f[0], f[1]) for f in feats]
    chs = [f[2] for f in feats]
    while len(hw) < cfg.num_levels:
        hw.append((_out(hw[-1][0], 2), _out(hw[-1][1], 2)))
    # extra levels
    L.append(('fpn.resample.3.conv', hw[2][0] * hw[2][1] * chs[2] * F, (hw[2][0] * hw[2][1] * (chs[2] + F) + chs[2] * F) * es, 'fpn.r3', hw[2][0] * hw[2][1] * F * es))
    L.append(('fpn.resample.3.pool', 0, (hw[2][0] * hw[2][1] + hw[3][0] * hw[3][1]) * F * es, 'fpn.r3p', hw[3][0] * hw[3][1] * F * es))
    L.append(('fpn.resample.4.pool', 0, (hw[3][0] * hw[3][1] + hw[4][0] * hw[4][1]) * F * es, 'fpn.r4p', hw[4][0] * hw[4][1] * F * es))
    nodes = get_fpn_config(cfg.fpn_name, cfg.min_level, cfg.max_level).nodes
    red0 = 2 ** cfg.min_level
    for ci in range(cfg.fpn_cell_repeats):
        for ni, node in enumerate(nodes):
            lvl = int(round(math.log2(node['reduction'] / red0)))
            px = hw[lvl][0] * hw[lvl][1]
            p = 'fpn.cell.%d.fnode.%d' % (ci, ni)
            in_px = 0
            for off in node['inputs_offsets']:
                if ci == 0 and off < 3:
                    spx = hw[off][0] * hw[off][1]
                    L.append(('%s.lateral.%d' % (p, off), spx * chs[off] * F, (spx * (chs[off] + F) + chs[off] * F) * es, p + '.lat%d' % off, spx * F * es))
                src_lvl = off if off < cfg.num_levels else int(round(math.log2(nodes[off - cfg.num_levels]['reduction'] / red0)))
                in_px += hw[src_lvl][0] * hw[src_lvl][1]
            # unfused: resample + fuse (n_in reads of the level + 1 write), act, dw (r+w), pw+BN (r+w)
            L.append((p + '.combine+act', 0, (in_px + px) * F * es, p, px * F * es))
            L.append((p + '.conv_dw', 9 * px * F, 2 * px * F * es, p, px * F * es))
            L.append((p + '.conv_pw', px * F * F, (2 * px * F + F * F) * es, p, px * F * es))
    P = sum(a * b for a, b in hw)
    for head, K in (('class_net', classes), ('box_net', 4)):
        for r in range(cfg.box_class_repeats):
            p = '%s.conv_rep.%d' % (head, r)
            L.append((p + '.conv_dw', 9 * P * F, 2 * P * F * es, p, P * F * es))
            L.append((p + '.conv_pw', P * F * F, (2 * P * F + F * F) * es, p, P * F * es))
        p = head + '.predict'
        L.append((p + '.conv_dw', 9 * P * F, 2 * P * F * es, p, P * F * es))
        L.append((p + '.conv_pw', P * F * A * K, (P * (F + A * K) + F * A * K) * es, p, P * A * K * es))
    return L, dict(P=P, N=A * P, F=F)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='tf_efficientdet_d0')
    ap.add_argument('--image', type=int, default=640)
    ap.add_argument('--classes', type=int, default=90)
    ap.add_argument('--es', type=int, default=2, help='bytes per activation element (2 = bf16)')
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--quiet', action='store_true')
    a = ap.parse_args()
    L, geo = layers(a.model, a.image, a.classes, a.es)
    if not a.quiet:
        print('%-52s %12s %12s %10s' % ('layer', 'MMAC/img', 'MB/img', 'FLOP/B'))
        for name, macs, nbytes, _, _ in L:
            print('%-52s %12.3f %12.3f %10.1f' % (name, macs / 1e6, nbytes / 1e6, 2 * macs / max(nbytes, 1)))
    tot_m = sum(l[1] for l in L)
    tot_b = sum(l[2] for l in L)
    t_layer = sum(max(2 * l[1] / MFMA, l[2] / HBM) for l in L)

    def part(prefix):
        return sum(l[1] for l in L if l[0].startswith(prefix)) / 1e9, sum(l[2] for l in L if l[0].startswith(prefix)) / 1e6
    print('\n%s %dx%d C=%d es=%d: N=%d anchors, P=%d pyramid pixels' % (a.model, a.image, a.image, a.classes, a.es, geo['N'], geo['P']))
    print('total  %.3f GMAC = %.2f GFLOP, %.1f MB per image (per-layer compulsory traffic), %.1f FLOP/B' % (
        tot_m / 1e9, 2 * tot_m / 1e9, tot_b / 1e6, 2 * tot_m / tot_b))
    for pre in ('backbone', 'fpn', 'class_net', 'box_net'):
        m, b = part(pre)
        print('  %-10s %.3f GMAC  %.1f MB' % (pre, m, b))
    print('t_roof (per layer)  = %.4f ms / image -> %.0f img/s ceiling; %.3f ms per %d-image step' % (
        1e3 * t_layer, 1.0 / t_layer, 1e3 * t_layer * a.batch, a.batch))
    # fused launch list: inside a group every member's output except the last stays on chip, i.e. its write and the next
    # member's read of it (2 * out_bytes) disappear from the HBM traffic
    groups, order = {}, []
    for name, macs, nbytes, g, ob in L:
        if g not in groups:
            groups[g] = [0, 0, []]
            order.append(g)
        groups[g][0] += macs
        groups[g][1] += nbytes
        groups[g][2].append(ob)
    t_fused, b_fused = 0.0, 0
    for g in order:
        macs, nbytes, obs = groups[g]
        gb = nbytes - 2 * sum(obs[:-1])
        b_fused += gb
        t_fused += max(2 * macs / MFMA, gb / HBM)
    print('t_roof (fused list) = %.4f ms / image -> %.0f img/s ceiling; %.1f MB per image, %d launches; %.3f ms per %d-image step' % (
        1e3 * t_fused, 1.0 / t_fused, b_fused / 1e6, len(order), 1e3 * t_fused * a.batch, a.batch))
    print("note: bench.py reports the fused bound from the engine's own launch metadata (roofline.t_roof_ms); this table is the"
          ' independent, weight-free restatement')


if __name__ == '__main__':
    main()
