"""Device-side input normalisation (reference: `PrefetchLoader`, effdet/data/loader.py:103-146).

The reference's loader hands the model `(uint8 - 255*mean) / (255*std)` computed with torch ops on the GPU
(loader.py:114-128).  Here the same arithmetic is one HIP kernel (`normalize_batch`), or - when a raw uint8
batch is passed straight to `EfficientDet.forward` / `DetBenchPredict.forward` - part of the network's first
kernel (`effdet_stem_dw_fused_u8`), so the normalised tensor is never written to memory.
`EfficientDet.input_mean / input_std` (ImageNet constants by default, `effdet/data/transforms.py:11-12`) are
the constants the fused path uses."""
import ctypes

import torch

from .. import _lib

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)


def normalize_batch(x: torch.Tensor, mean=IMAGENET_DEFAULT_MEAN, std=IMAGENET_DEFAULT_STD, dtype=torch.float32):
    """uint8 [B, C, H, W] on a GPU -> `(x - 255*mean) / (255*std)` as `dtype` (float32 or bfloat16)."""
    if x.dtype != torch.uint8 or x.dim() != 4:
        raise ValueError('expected a uint8 [B, C, H, W] tensor')
    if x.device.type != 'cuda':
        raise RuntimeError('normalize_batch runs on the GPU only (no CPU fallback)')
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError('dtype must be float32 or bfloat16')
    B, C, H, W = x.shape
    if len(mean) != C or len(std) != C or C > 4:
        raise ValueError('mean / std must have one entry per channel (at most 4 channels)')
    lib = _lib.load()
    x = x.contiguous()
    y = torch.empty(B, C, H, W, dtype=dtype, device=x.device)
    m = (ctypes.c_float * C)(*[255.0 * v for v in mean])
    s = (ctypes.c_float * C)(*[255.0 * v for v in std])
    st = torch.cuda.current_stream(x.device).cuda_stream
    _lib.check(lib.effdet_normalize_u8(st, 0 if dtype == torch.float32 else 1, x.data_ptr(), m, s, y.data_ptr(), B, C, H * W),
               'effdet_normalize_u8')
    return y
