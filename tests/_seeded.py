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
