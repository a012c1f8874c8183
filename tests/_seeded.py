"""Deterministic synthetic tensors shared by tools/make_golden.py and the tests.

numpy's legacy RandomState(seed) stream is stable across numpy versions, so a fixture only has to
store (key, shape) pairs and the seed - not megabytes of weights.
"""
import zlib

import numpy as np
import torch


def _rs(seed, key):
    return np.random.RandomState((seed * 1000003 + zlib.crc32(key.encode())) % (2 ** 31 - 1))


def seeded_tensor(seed, key, shape):
    """Value distribution chosen by the key's role so folding / ordering bugs are visible."""
    rs = _rs(seed, key)
    shape = tuple(int(s) for s in shape)
    if key.endswith('running_var'):
        a = rs.uniform(0.5, 1.5, shape)
    elif key.endswith('running_mean'):
        a = rs.normal(0.0, 0.1, shape)
    elif key.endswith('num_batches_tracked'):
        a = np.zeros(shape)
    elif key.endswith('edge_weights'):
        a = rs.uniform(-0.3, 2.0, shape)          # some negatives: relu in fastattn matters
    elif '.bn' in key and key.endswith('weight'):
        a = rs.uniform(0.5, 1.5, shape)
    elif key.endswith('bias'):
        a = rs.normal(0.0, 0.1, shape)
    elif len(shape) == 4:
        fan_in = shape[1] * shape[2] * shape[3]
        a = rs.normal(0.0, 1.0, shape) * (1.7 / np.sqrt(fan_in))
    else:
        a = rs.normal(0.0, 1.0, shape)
    return torch.from_numpy(np.asarray(a, dtype=np.float32))


def seeded_state_dict(seed, keys, shapes):
    return {k: seeded_tensor(seed, k, s) for k, s in zip(keys, shapes)}


def seeded_array(seed, key, shape, kind='normal', scale=1.0):
    rs = _rs(seed, key)
    if kind == 'normal':
        return (rs.normal(0.0, scale, shape)).astype(np.float32)
    if kind == 'uniform':
        return (rs.uniform(0.0, scale, shape)).astype(np.float32)
    raise ValueError(kind)


def eval_case(seed, n_img, C, n_det):
    """seeded evaluation inputs shared with the tests: per image ground truth (1..5 boxes) and detections = jittered copies of
    ground-truth boxes + random boxes + a few degenerate (invalid) ones, distinct scores (no ties), sorted descending."""
    rs = np.random.RandomState(seed)
    images = []
    for _ in range(n_img):
        m = rs.randint(0, 6)
        y0 = rs.uniform(0, 80, m); x0 = rs.uniform(0, 80, m)
        gt = np.stack([y0, x0, y0 + rs.uniform(8, 60, m), x0 + rs.uniform(8, 60, m)], 1).astype(np.float32).reshape(-1, 4)
        gc = rs.randint(1, C + 1, m)
        det, dc = [], []
        for _k in range(n_det):
            r = rs.uniform()
            if m > 0 and r < 0.6:
                j = rs.randint(0, m)
                det.append(gt[j] + rs.normal(0, 3.0, 4).astype(np.float32))
                dc.append(gc[j] if rs.uniform() < 0.8 else rs.randint(1, C + 1))
            elif r < 0.95:
                yy, xx = rs.uniform(0, 80), rs.uniform(0, 80)
                det.append(np.array([yy, xx, yy + rs.uniform(5, 50), xx + rs.uniform(5, 50)], np.float32))
                dc.append(rs.randint(1, C + 1))
            else:
                yy, xx = rs.uniform(0, 80), rs.uniform(0, 80)
                det.append(np.array([yy, xx, yy - 1.0, xx + 5.0], np.float32))           # invalid: ymax < ymin
                dc.append(rs.randint(1, C + 1))
        sc = np.sort(rs.permutation(10000)[:n_det].astype(np.float32) / 10000.0)[::-1].copy()
        images.append(dict(gt_boxes=gt, gt_classes=np.asarray(gc, np.int64), det_boxes=np.stack(det).astype(np.float32),
                           det_scores=sc, det_classes=np.asarray(dc, np.int64)))
    return images


def meta_nets_case(g):
    """Rebuild the seeded weights / inputs of tests/golden/meta_nets.npz (tools/make_golden.py::gen_meta_nets): the
    fixture stores the reference's outputs only, the inputs are regenerated from (seed, key, shape)."""
    seed, B, Fc, A, L, R = [int(v) for v in g['mh_meta'][:6]]
    sizes = [int(v) for v in g['mh_meta'][6:]]
    init = {}
    for l in range(R):
        init['class_net.conv_rep.%d.conv_dw.weight' % l] = seeded_tensor(seed, 'class_net.conv_rep.%d.conv_dw.weight' % l, (Fc, 1, 3, 3))
        init['class_net.conv_rep.%d.conv_pw.weight' % l] = seeded_tensor(seed, 'class_net.conv_rep.%d.conv_pw.weight' % l, (Fc, Fc, 1, 1))
        init['class_net.conv_rep.%d.conv_pw.bias' % l] = seeded_tensor(seed, 'class_net.conv_rep.%d.conv_pw.bias' % l, (Fc,))
        for lev in range(L):
            for wb in ('weight', 'bias'):
                k = 'class_net.bn_rep.%d.%d.bn.%s' % (l, lev, wb)
                init[k] = seeded_tensor(seed, k, (Fc,))
    init['class_net.predict.conv_dw.weight'] = seeded_tensor(seed, 'class_net.predict.conv_dw.weight', (Fc, 1, 3, 3))
    sc = (1.0 / Fc) ** 0.5
    extra = dict(predict_pw=seeded_tensor(seed, 'meta.predict_pw', (A, Fc, 1, 1)) * sc,
                 predict_pb=seeded_tensor(seed, 'meta.predict_pb', (A,)),
                 predict_pw_sep=seeded_tensor(seed, 'meta.predict_pw_sep', (A, Fc, 1, 1)) * sc,
                 predict_pb_sep=seeded_tensor(seed, 'meta.predict_pb_sep', (A,)))
    x = [torch.from_numpy(seeded_array(seed, 'lvl%d' % i, (B, Fc, s, s))) for i, s in enumerate(sizes)]
    xa = [torch.from_numpy(seeded_array(seed + 1, 'an%d' % i, (2, Fc, s, s))) for i, s in enumerate(sizes)]
    xp = torch.from_numpy(seeded_array(seed + 2, 'px', (5, 37, Fc + 42)))
    return dict(seed=seed, B=B, F=Fc, A=A, L=L, R=R, sizes=sizes, init=init, extra=extra, x=x, x_anchor=xa, x_proj=xp)


def meta_lists(init, extra, L, R):
    """the reference's parameter lists (efficientdet.py:594-632): conv_dw, conv_pw, conv_pb, predict, bn_w, bn_b (level-major)"""
    dw = [init['class_net.conv_rep.%d.conv_dw.weight' % l] for l in range(R)]
    pw = [init['class_net.conv_rep.%d.conv_pw.weight' % l] for l in range(R)]
    pb = [init['class_net.conv_rep.%d.conv_pw.bias' % l] for l in range(R)]
    pred = [init['class_net.predict.conv_dw.weight'], extra['predict_pw'], extra['predict_pb']]
    bw = [init['class_net.bn_rep.%d.%d.bn.weight' % (r, lev)] for lev in range(L) for r in range(R)]
    bb = [init['class_net.bn_rep.%d.%d.bn.bias' % (r, lev)] for lev in range(L) for r in range(R)]
    return dw, pw, pb, pred, bw, bb
