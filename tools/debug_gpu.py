import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch
from _models import seeded_model
from _seeded import seeded_array
from oracle import model as om
from oracle import postprocess as op
from ood_object_detection_amd.effdet.bench import DetBenchPredict, _post_process

DEV = 'cuda:0'
model, cfg, nodes, sd = seeded_model('tf_efficientdet_d0', 256, 90, seed=3)
x = torch.from_numpy(seeded_array(3, 'input', (2, 3, 256, 256)))
with torch.no_grad():
    feats, activs = om.efficientdet_forward(sd, cfg, x, nodes, mode='fpn')
    cls_r, box_r = om.efficientdet_forward(sd, cfg, x, nodes)

def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).abs().max()), float(b.abs().max()), float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())

for dtype in (torch.float32, torch.bfloat16):
    m = model.to(DEV).to(dtype)
    xd = x.to(DEV).to(dtype)
    f = m(xd, mode='bb')
    print(dtype, 'feats (linf, max|ref|, rel-rms):', [rel(a, b) for a, b in zip(f, feats)])
    _, a = m(xd, mode='fpn')
    print(dtype, 'activs', [rel(p, q) for p, q in zip(a, activs)])
    c, b = m(xd)
    print(dtype, 'cls', [rel(p, q) for p, q in zip(c, cls_r)])
    print(dtype, 'box', [rel(p, q) for p, q in zip(b, box_r)])

m = model.to(DEV).float()
c, b = m(x.to(DEV))
gc, gb, gi, gcl = _post_process(c, b, 5, 90, 5000)
rc, rb, ri, rcl = op.post_process(cls_r, box_r, 5, 90, 5000)
print('topk gpu logits[:8]', gc[0, :8, 0].tolist())
print('topk ref logits[:8]', rc[0, :8, 0].tolist())
print('cls gpu', gcl[0, :8].tolist(), 'ref', rcl[0, :8].tolist())
print('idx gpu', gi[0, :8].tolist(), 'ref', ri[0, :8].tolist())
print('n index mismatches', int((gi.cpu() != ri).sum()), 'class mismatches', int((gcl.cpu() != rcl).sum()))
bench = DetBenchPredict(m).to(DEV)
det = bench(x.to(DEV))
anchors = op.anchor_boxes(3, 7, 3, cfg.aspect_ratios, 4.0, (256, 256))
ref, src = op.generate_detections(rc[0], rb[0], anchors, ri[0], rcl[0], None, torch.tensor(256), 100, False, return_aux=True)
print('det gpu[:6]', det[0, :6].tolist())
print('det ref[:6]', ref[:6].tolist())
print('counts', bench.last_count.tolist(), ref.shape)
