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


# ---- ResizePad on the device (effdet/data/transforms.py:75-107) --------------------------------------
_PRECISION_BITS = 32 - 8 - 2


def _pil_bilinear_tables(in_size, out_size):
    """Pillow's BILINEAR coefficient tables (Resample.c precompute_coeffs + normalize_coeffs_8bpc), computed in
    float64 with Pillow's operation order: (bounds int32 [out,2], coefficients int32 [out,ksize])."""
    import numpy as np
    if in_size == out_size:                                   # Pillow skips the pass; identity coefficients do the same
        b = np.stack([np.arange(out_size), np.ones(out_size, np.int64)], 1).astype(np.int32)
        return b, np.full((out_size, 1), 1 << _PRECISION_BITS, np.int32)
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    xx = np.arange(out_size, dtype=np.float64)
    center = (xx + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)           # C cast: truncation toward zero
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    x = np.arange(ksize, dtype=np.int64)[None, :]
    a = np.abs((x + xmin[:, None] - center[:, None] + 0.5) * ss)
    w = np.where((a < 1.0) & (x < xmax[:, None]), 1.0 - a, 0.0)
    ww = np.zeros(out_size, np.float64)
    for j in range(ksize):                                     # Pillow sums the weights left to right
        ww = ww + w[:, j]
    k = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    kk = np.where(k < 0, (-0.5 + k * (1 << _PRECISION_BITS)).astype(np.int64), (0.5 + k * (1 << _PRECISION_BITS)).astype(np.int64))
    kk = np.where(x < xmax[:, None], kk, 0)
    return np.stack([xmin, xmax], 1).astype(np.int32), kk.astype(np.int32)


def resolve_fill_color(fill_color='mean', img_mean=IMAGENET_DEFAULT_MEAN):
    """effdet/data/transforms.py:279-290."""
    if isinstance(fill_color, tuple):
        assert len(fill_color) == 3
        return fill_color
    try:
        return (int(fill_color),) * 3
    except ValueError:
        assert fill_color == 'mean'
        return tuple(int(round(255 * x)) for x in img_mean)


def resize_pad(img: torch.Tensor, target_size: int, fill_color=(0, 0, 0)):
    """`ResizePad.__call__` for one image: uint8 [h, w, 3] GPU tensor -> (uint8 [3, S, S] ready for the batch,
    img_scale = 1 / scale as stored in anno['img_scale'])."""
    import numpy as np
    if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
        raise ValueError('expected a uint8 [h, w, 3] image')
    if img.device.type != 'cuda':
        raise RuntimeError('resize_pad runs on the GPU only (no CPU fallback)')
    h, w = int(img.shape[0]), int(img.shape[1])
    S = int(target_size)
    img_scale = min(S / h, S / w)
    sh, sw = int(h * img_scale), int(w * img_scale)
    lib = _lib.load()
    dev = img.device
    bx, kx = _pil_bilinear_tables(w, sw)
    by, ky = _pil_bilinear_tables(h, sh)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    bxd, kxd, byd, kyd = t(bx), t(kx), t(by), t(ky)
    img = img.contiguous()
    out = torch.empty(3, S, S, dtype=torch.uint8, device=dev)
    ws = torch.empty(h * sw * 3, dtype=torch.uint8, device=dev)
    fill = (ctypes.c_int * 3)(*[int(v) for v in fill_color])
    st = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(lib.effdet_resize_pad_u8(st, img.data_ptr(), h, w, out.data_ptr(), S, sw, sh, bxd.data_ptr(), kxd.data_ptr(), kx.shape[1],
                                        byd.data_ptr(), kyd.data_ptr(), ky.shape[1], fill, ws.data_ptr()), 'effdet_resize_pad_u8')
    return out, 1.0 / img_scale
