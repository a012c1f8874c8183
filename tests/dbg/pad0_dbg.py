import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from _models import seeded_model
from _seeded import seeded_array
from oracle import model as om
DEV = 'cuda:0'
for name in ('efficientdet_d0', 'efficientdet_d1', 'tf_efficientdet_d1'):
    model, cfg, nodes, sd = seeded_model(name, 128, 3, seed=5)
    x = torch.from_numpy(seeded_array(6, 'input', (2, 3, 128, 128)))
    with torch.no_grad():
        fr = om.backbone_forward(sd, cfg.backbone_name, x)
        info = om.backbone_feature_info(cfg.backbone_name)
        ar = om.bifpn_forward(sd, cfg, fr, nodes, info)
    m = model.to(DEV).float()
    with torch.no_grad():
        feats, activs = m(x.to(DEV), mode='fpn')
    print(name, 'feat err', [round(float((a.cpu() - b).abs().max() / b.abs().max()), 6) for a, b in zip(feats, fr)],
          'act err', [round(float((a.cpu() - b).abs().max() / b.abs().max()), 6) for a, b in zip(activs, ar)])
