"""Stub of timm.models.layers: conv/pool factories with 'same' and '' padding, Swish."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle.model import conv2d_pad, maxpool_pad


class _Conv2dPad(nn.Conv2d):
    def __init__(self, cin, cout, k, stride=1, dilation=1, groups=1, bias=True, pad_type=''):
        super().__init__(cin, cout, k, stride=stride, padding=0, dilation=dilation, groups=groups, bias=bias)
        assert dilation == 1
        self.pad_type = pad_type

    def forward(self, x):
        return conv2d_pad(x, self.weight, self.bias, self.stride[0], self.pad_type, self.groups)


def create_conv2d(in_channels, out_channels, kernel_size, **kwargs):
    depthwise = kwargs.pop('depthwise', False)
    padding = kwargs.pop('padding', '')
    groups = out_channels if depthwise else kwargs.pop('groups', 1)
    return _Conv2dPad(in_channels, out_channels, kernel_size, stride=kwargs.pop('stride', 1),
                      dilation=kwargs.pop('dilation', 1), groups=groups, bias=kwargs.pop('bias', False),
                      pad_type=padding)


class _MaxPoolPad(nn.Module):
    def __init__(self, k, s, pad_type):
        super().__init__()
        self.k, self.s, self.pad_type = k, s, pad_type

    def forward(self, x):
        return maxpool_pad(x, self.k, self.s, self.pad_type)


def create_pool2d(pool_type, kernel_size, stride=None, **kwargs):
    assert pool_type == 'max'
    return _MaxPoolPad(kernel_size, stride or kernel_size, kwargs.pop('padding', ''))


class Swish(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()

    def forward(self, x):
        return x * torch.sigmoid(x)


def get_act_layer(name='relu'):
    if not name:
        return None
    return {'swish': Swish, 'silu': Swish, 'relu': nn.ReLU}[name]
