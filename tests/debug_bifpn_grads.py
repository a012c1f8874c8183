"""Debug aid: per-tensor / per-node gradient comparison of the BiFPN + heads backward (HIP TrainEngine vs autograd through the
oracle), incl. max-pool near-tie statistics.  usage: python tests/debug_bifpn_grads.py [bifpn_fa|bifpn_attn|bifpn_sum]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np, torch
from _models import seeded_model
from _seeded import seeded_array
from oracle import model as om
from ood_object_detection_amd.train_engine import TrainEngine
fpn_name = sys.argv[1] if len(sys.argv) > 1 else 'bifpn_sum'
size, B, C = 128, 2, 12
model, cfg, nodes, sd = seeded_model('tf_efficientdet_d0', size, C, seed=37, fpn_name=fpn_name)
x = torch.from_numpy(seeded_array(37, 'input', (B, 3, size, size)))
sdf = {k: v.float() for k, v in sd.items()}
info = om.backbone_feature_info(cfg.backbone_name)
with torch.no_grad():
    feats = om.backbone_forward(sdf, cfg.backbone_name, x, pad_type=cfg.pad_type)
feats = [f.clone().requires_grad_() for f in feats]
# oracle BiFPN with every node output retained
eps = cfg.norm_kwargs['eps']
xs = list(feats); inf = [dict(f) for f in info]
for level in range(cfg.num_levels):
    if level < len(inf): continue
    xs.append(om._resample(xs[-1], sdf, 'fpn.resample.%d.' % level, cfg, inf[-1]['num_chs'], 2, eps))
    inf.append(dict(num_chs=cfg.fpn_channels, reduction=inf[-1]['reduction'] * 2))
allx = list(xs); ids = list(range(len(xs))); fused_list = []; act_list = []
for rep in range(cfg.fpn_cell_repeats):
    for ni, node in enumerate(nodes):
        p = 'fpn.cell.%d.fnode.%d.' % (rep, ni)
        ins = []
        for off in node['inputs_offsets']:
            if off < len(inf): in_chs, in_red = inf[off]['num_chs'], inf[off]['reduction']
            else: in_chs, in_red = cfg.fpn_channels, nodes[off - len(inf)]['reduction']
            ins.append(om._resample(allx[ids[off]], sdf, '%scombine.resample.%d.' % (p, off), cfg, in_chs, node['reduction'] / in_red, eps))
        y = om._combine(ins, sdf.get(p + 'combine.edge_weights'), node['weight_method'])
        y.retain_grad(); fused_list.append(y)
        y = om.silu(y)
        y.retain_grad(); act_list.append(y)
        y = om._sepconv(y, sdf, p + 'after_combine.conv.', cfg.pad_type, bn_eps=eps, act=False)
        y.retain_grad()
        allx.append(y); ids.append(len(allx) - 1)
    ids = ids[-cfg.num_levels:]
    inf = [dict(num_chs=cfg.fpn_channels, reduction=n['reduction']) for n in nodes[-cfg.num_levels:]]
for t in allx[3:5]: t.retain_grad()
pyr = [allx[i] for i in ids]
co = om.head_forward(sdf, cfg, pyr, 'class_net.'); bo = om.head_forward(sdf, cfg, pyr, 'box_net.')
g = torch.Generator().manual_seed(1)
gc = [torch.randn(t.shape, generator=g) for t in co]; gb = [torch.randn(t.shape, generator=g) for t in bo]
torch.autograd.backward(co + bo, gc + gb)
model = model.to('cuda:0').float().train()
model.apply(lambda m: m.eval() if isinstance(m, torch.nn.BatchNorm2d) else None)
eng = TrainEngine(model); eng.keep_debug = True
fd = [f.detach().permute(0, 2, 3, 1).contiguous().to('cuda:0') for f in feats]
cls_all, box_all, saved = eng.fh_forward(fd)
Bn = cls_all.shape[0]
gcp = torch.cat([t.permute(0, 2, 3, 1).reshape(Bn, -1, C) for t in gc], 1).contiguous().to('cuda:0')
gbp = torch.cat([t.permute(0, 2, 3, 1).reshape(Bn, -1, 4) for t in gb], 1).contiguous().to('cuda:0')
dfeats, grads = eng.fh_backward(gcp, gbp, saved)
dt = eng._debug_dt
for i, t in enumerate(allx):
    if t.grad is None or dt[i] is None: print(i, 'none'); continue
    ref = t.grad.permute(0, 2, 3, 1)
    err = float((dt[i].cpu() - ref).abs().max() / ref.abs().max())
    print('tensor %2d  shape %-18s rel err %.3e  |ref| %.3e' % (i, tuple(ref.shape), err, float(ref.abs().max())))

for k, nrec in enumerate(saved['nodes']):
    rf = fused_list[k].grad.permute(0, 2, 3, 1); ra = act_list[k].grad.permute(0, 2, 3, 1)
    ef = float((nrec['_dfused'].cpu() - rf).abs().max() / rf.abs().max())
    ea = float((nrec['_dact'].cpu() - ra).abs().max() / ra.abs().max())
    fz = float((nrec['fused'].cpu() - fused_list[k].detach().permute(0, 2, 3, 1)).abs().max() / fused_list[k].abs().max())
    print('node %2d (tensor %2d) n_in %d  err dact %.2e  err dfused %.2e  err fused(saved) %.2e  |fused| %.1f' % (k, nrec['out_id'], nrec['n_in'], ea, ef, fz, float(fused_list[k].abs().max())))
n16, n22 = saved['nodes'][16], saved['nodes'][22]
hip_sum = (n16['_dfused'] + n22['_dfused']).cpu()
ref19 = allx[19].grad.permute(0, 2, 3, 1)
ref_sum = (fused_list[16].grad + fused_list[22].grad).permute(0, 2, 3, 1)
print('HIP  dt[19] vs HIP dfused21+dfused27 :', float((dt[19].cpu() - hip_sum).abs().max()))
print('ORACLE d19  vs ORACLE df21+df27      :', float((ref19 - ref_sum).abs().max()), ' |d19|', float(ref19.abs().max()))
print('node16 src_ids', n16['src_ids'], 'node22 src_ids', n22['src_ids'], 'w', n16['w'].tolist(), n22['w'].tolist(), float(n16['den']))
for i in (16, 17, 18, 19):
    ref = allx[i].grad.permute(0, 2, 3, 1)
    d = (dt[i].cpu() - ref).abs()
    thr = 1e-3 * float(ref.abs().max())
    print('tensor %d: %d of %d entries off by > 1e-3 max; worst %.3e' % (i, int((d > thr).sum()), d.numel(), float(d.max())))
# near-ties in the pooled tensors (cell 1): top-2 gap of every 3x3/s2 SAME window
import torch.nn.functional as F
for i in (17, 18, 19):
    x_ = allx[i].detach()
    pt, pb = om.same_pad_amounts(x_.shape[-2], 3, 2); pl, pr = om.same_pad_amounts(x_.shape[-1], 3, 2)
    xp = F.pad(x_, [pl, pr, pt, pb], value=float('-inf'))
    win = F.unfold(xp, 3, stride=2).reshape(x_.shape[0], x_.shape[1], 9, -1)
    top2 = win.topk(2, dim=2).values
    gap = (top2[:, :, 0] - top2[:, :, 1]).abs()
    print('tensor %d pooled: windows %d, gap < 1e-5: %d, gap == 0: %d' % (i, gap.numel(), int((gap < 1e-5).sum()), int((gap == 0).sum())))
