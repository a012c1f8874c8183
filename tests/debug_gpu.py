import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch
from _models import seeded_model
from _seeded import seeded_array
from oracle import model as om
DEV = 'cuda:0'
model, cfg, nodes, sd = seeded_model('tf_efficientdet_d0', 128, 20, seed=11)
x = torch.from_numpy(seeded_array(11, 'input', (2, 3, 128, 128)))
with torch.no_grad():
    cls_r, box_r = om.efficientdet_forward(sd, cfg, x, nodes)
m = model.to(DEV).float()
for it in range(3):
    c, b = m(x.to(DEV))
    torch.cuda.synchronize()
    for name, got, ref in (('cls', c, cls_r), ('box', b, box_r)):
        for l, (g, r) in enumerate(zip(got, ref)):
            g = g.float().cpu()
            bad = ~torch.isfinite(g)
            print(it, name, l, tuple(g.shape), 'nonfinite', int(bad.sum()), 'linf', float((torch.nan_to_num(g) - r).abs().max()))
            if bad.any():
                idx = bad.nonzero()[:5]
                print('   first bad idx', idx.tolist())
    e = m.ood_energy.cpu()
    print(it, 'energy nonfinite', int((~torch.isfinite(e)).sum()))
