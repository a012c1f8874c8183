"""ORACLE (test infrastructure only - never imported by the product path).

CPU restatement of the reference's input normalisation: `PrefetchLoader.__init__` / `__iter__`
(effdet/data/loader.py:114-115, 127-128): `mean = 255 * IMAGENET_DEFAULT_MEAN`, `std = 255 * IMAGENET_DEFAULT_STD`
as float32 `[1,3,1,1]` tensors, `input.float().sub_(mean).div_(std)`.  Parity pin: the arithmetic is three torch
ops; `tests/test_oracle_golden.py::test_normalize_matches_loader_expression` checks this restatement against the
literal expression (the loader class itself needs CUDA at construction and cannot be instantiated here)."""
import numpy as np

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)      # effdet/data/transforms.py:11
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)       # effdet/data/transforms.py:12


def normalize_u8(x, mean=IMAGENET_DEFAULT_MEAN, std=IMAGENET_DEFAULT_STD):
    """x: uint8 [B,C,H,W] numpy -> float32 [B,C,H,W], float32 arithmetic in the reference's order."""
    x = np.asarray(x)
    assert x.dtype == np.uint8 and x.ndim == 4
    m = np.array([v * 255 for v in mean], dtype=np.float32).reshape(1, -1, 1, 1)   # python-float product, then f32 (loader.py:114)
    s = np.array([v * 255 for v in std], dtype=np.float32).reshape(1, -1, 1, 1)
    return (x.astype(np.float32) - m) / s
