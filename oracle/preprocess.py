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


# ------------------------------------------------------------------------------------------------
# ResizePad (effdet/data/transforms.py:75-107): letterbox to target_size with PIL bilinear resize, paste top-left
# on a fill-colour canvas.  The resize arithmetic lives in Pillow (third-party, `Image.resize(.., BILINEAR)`,
# version 12.2.0 installed here); `pil_bilinear_coeffs` / `pil_resize_bilinear` restate its published 8-bit
# algorithm (src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal/
# Vertical_8bpc) and tests/test_oracle_golden.py::test_resize_pad_matches_pil pins them bit-exactly against PIL itself.
# ------------------------------------------------------------------------------------------------
PRECISION_BITS = 32 - 8 - 2


def pil_bilinear_coeffs(in_size, out_size):
    """-> (bounds int32 [out,2] = (xmin, count), coefficients int32 [out, ksize]) of Pillow's BILINEAR filter."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size) - xmin
        w = np.zeros(xmax, np.float64)
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - a if a < 1.0 else 0.0
        ww = 0.0
        for x in range(xmax):
            ww += w[x]
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis0(img, bounds, kk):
    """img uint8 [n, ...]: resample along axis 0 with the integer coefficients; uint8 out."""
    out = np.empty((bounds.shape[0],) + img.shape[1:], np.uint8)
    x64 = img.astype(np.int64)
    for i in range(bounds.shape[0]):
        xmin, cnt = int(bounds[i, 0]), int(bounds[i, 1])
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(cnt):
            acc += x64[xmin + x] * int(kk[i, x])
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def pil_resize_bilinear(img, out_w, out_h):
    """img uint8 [h, w, c] -> uint8 [out_h, out_w, c]; horizontal pass first, then vertical, each rounded to 8 bits
    (a pass whose size does not change is skipped, as in ImagingResample)."""
    h, w = img.shape[:2]
    if out_w != w:
        b, k = pil_bilinear_coeffs(w, out_w)
        img = np.transpose(_resample_axis0(np.transpose(img, (1, 0, 2)), b, k), (1, 0, 2))
    if out_h != h:
        b, k = pil_bilinear_coeffs(h, out_h)
        img = _resample_axis0(img, b, k)
    return img


def resolve_fill_color(fill_color='mean', img_mean=IMAGENET_DEFAULT_MEAN):      # transforms.py:279-290
    if isinstance(fill_color, tuple):
        return fill_color
    try:
        return (int(fill_color),) * 3
    except ValueError:
        return tuple(int(round(255 * x)) for x in img_mean)


def resize_pad(img, target_size, fill_color=(0, 0, 0)):
    """ResizePad.__call__ on a uint8 [h, w, 3] array -> (uint8 [S, S, 3], img_scale returned in anno = 1/scale)."""
    h, w = img.shape[:2]
    img_scale = min(target_size / h, target_size / w)
    sh, sw = int(h * img_scale), int(w * img_scale)
    canvas = np.empty((target_size, target_size, 3), np.uint8)
    canvas[:] = np.array(fill_color, np.uint8)
    canvas[:sh, :sw] = pil_resize_bilinear(img, sw, sh)
    return canvas, 1.0 / img_scale
