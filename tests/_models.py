"""Shared model builders for tests / smoke / bench: seeded weights for product model + oracle."""
import torch

from _seeded import seeded_tensor


def seeded_model(name='tf_efficientdet_d0', image_size=256, num_classes=90, seed=3, cls_bias=None, soft_nms=False, fpn_name=None,
                 drop_path_rate=0.0):
    """drop_path_rate: the configs' default is 0.2 (stochastic depth in the TRAINING forward); the parity tests need a
    deterministic step and switch it off unless they test it (fixed masks: backbone.drop_path_masks)"""
    from ood_object_detection_amd.effdet.config import get_efficientdet_config, get_fpn_config
    from ood_object_detection_amd.effdet.efficientdet import EfficientDet
    cfg = get_efficientdet_config(name)
    cfg.image_size = (image_size, image_size)
    cfg.num_classes = num_classes
    cfg.soft_nms = soft_nms
    cfg.backbone_args = dict(drop_path_rate=drop_path_rate)
    if fpn_name is not None:
        cfg.fpn_name = fpn_name
    model = EfficientDet(cfg, pretrained_backbone=False).eval()
    sd = model.state_dict()
    new = {}
    for k, v in sd.items():
        new[k] = v if k.endswith('num_batches_tracked') else seeded_tensor(seed, k, v.shape)
    # keep head outputs in a realistic range: class logits O(1), box regressions O(0.3)
    new['class_net.predict.conv_pw.weight'] = new['class_net.predict.conv_pw.weight'] * 0.1
    new['box_net.predict.conv_pw.weight'] = new['box_net.predict.conv_pw.weight'] * 0.01
    if cls_bias is not None:
        new['class_net.predict.conv_pw.bias'] = torch.full_like(new['class_net.predict.conv_pw.bias'], cls_bias)
    # one oracle pass on seeded images sets every BN's running stats to what it actually sees
    from oracle import model as om
    from _seeded import seeded_array
    nodes = get_fpn_config(cfg.fpn_name, cfg.min_level, cfg.max_level).nodes
    xcal = torch.from_numpy(seeded_array(seed + 1000, 'calib', (2, 3, image_size, image_size)))
    om.CALIBRATE = True
    try:
        with torch.no_grad():
            om.efficientdet_forward(new, cfg, xcal, nodes)
    finally:
        om.CALIBRATE = False
    # tiny maps (P6/P7 of a small image) give 2-8 samples per channel: keep their variances sane
    for k in list(new.keys()):
        if k.endswith('running_var'):
            new[k] = new[k].clamp(min=0.25)
    model.load_state_dict(new, strict=True)
    nodes = get_fpn_config(cfg.fpn_name, cfg.min_level, cfg.max_level).nodes
    return model, cfg, nodes, {k: v.clone() for k, v in new.items()}
