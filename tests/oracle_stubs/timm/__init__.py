"""Stub of timm: create_model returns the oracle's EfficientNet feature extractor."""
import torch
import torch.nn as nn

from oracle import model as _om


class _FeatureInfo(object):
    def __init__(self, info):
        self.info = info

    def get_dicts(self, keys=None):
        return [{k: d[k] for k in (keys or d.keys())} for d in self.info]


class _OracleBackbone(nn.Module):
    def __init__(self, name):
        super().__init__()
        self.name = name
        self.feature_info = _FeatureInfo(_om.backbone_feature_info(name))
        stem, stages = _om.efficientnet_spec(name)
        shapes = {'conv_stem.weight': (stem, 3, 3, 3)}
        _bn(shapes, 'bn1', stem)
        for si, blocks in enumerate(stages):
            for bi, b in enumerate(blocks):
                p = 'blocks.%d.%d.' % (si, bi)
                if b['type'] == 'ds':
                    shapes[p + 'conv_dw.weight'] = (b['cin'], 1, b['k'], b['k'])
                    _bn(shapes, p + 'bn1', b['cin'])
                    _se(shapes, p + 'se.', b['cin'], b['se'])
                    shapes[p + 'conv_pw.weight'] = (b['cout'], b['cin'], 1, 1)
                    _bn(shapes, p + 'bn2', b['cout'])
                else:
                    shapes[p + 'conv_pw.weight'] = (b['mid'], b['cin'], 1, 1)
                    _bn(shapes, p + 'bn1', b['mid'])
                    shapes[p + 'conv_dw.weight'] = (b['mid'], 1, b['k'], b['k'])
                    _bn(shapes, p + 'bn2', b['mid'])
                    _se(shapes, p + 'se.', b['mid'], b['se'])
                    shapes[p + 'conv_pwl.weight'] = (b['cout'], b['mid'], 1, 1)
                    _bn(shapes, p + 'bn3', b['cout'])
        self._keys = list(shapes)
        for k, s in shapes.items():
            t = torch.zeros(s)
            if k.endswith('running_mean') or k.endswith('running_var') or k.endswith('num_batches_tracked'):
                self.register_buffer(k.replace('.', '__'), t)
            else:
                self.register_parameter(k.replace('.', '__'), nn.Parameter(t))

    def state_dict(self, destination=None, prefix='', keep_vars=False):
        out = destination if destination is not None else {}
        for k in self._keys:
            t = getattr(self, k.replace('.', '__'))
            out[prefix + k] = t if keep_vars else t.detach()
        return out

    def _load_from_state_dict(self, state_dict, prefix, *a, **k):
        for key in self._keys:
            getattr(self, key.replace('.', '__')).data.copy_(state_dict[prefix + key])

    def forward(self, x):
        sd = {'backbone.' + k: getattr(self, k.replace('.', '__')) for k in self._keys}
        return _om.backbone_forward(sd, self.name, x)


def _bn(shapes, p, c):
    for s in ('weight', 'bias', 'running_mean', 'running_var'):
        shapes['%s.%s' % (p, s)] = (c,)


def _se(shapes, p, c, r):
    shapes[p + 'conv_reduce.weight'] = (r, c, 1, 1)
    shapes[p + 'conv_reduce.bias'] = (r,)
    shapes[p + 'conv_expand.weight'] = (c, r, 1, 1)
    shapes[p + 'conv_expand.bias'] = (c,)


def create_model(name, features_only=True, out_indices=(2, 3, 4), pretrained=False, **kwargs):
    assert features_only and tuple(out_indices) == (2, 3, 4)
    return _OracleBackbone(name)
