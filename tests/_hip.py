"""Thin test-side helpers to call the C ABI with torch tensors (GPU tests only)."""
import ctypes

import torch

from ood_object_detection_amd import _lib

DT = {torch.float32: 0, torch.bfloat16: 1}


def stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def ptr(t):
    return None if t is None else t.data_ptr()


def nhwc(x, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(dtype)


def nchw(x):
    return x.permute(0, 3, 1, 2).float()


def fold_bn(w, b, m, v, eps, conv_bias=None):
    scale = w / torch.sqrt(v + eps)
    shift = b - m * scale
    if conv_bias is not None:
        shift = shift + conv_bias * scale
    return scale.float().contiguous(), shift.float().contiguous()


def pw_gemm(A, W, scale, shift, act=0, residual=None, gate=None, rows_per_image=0):
    lib = _lib.load()
    M, K = A.shape
    N = W.shape[0]
    C = torch.empty(M, N, dtype=A.dtype, device=A.device)
    rc = lib.effdet_pw_gemm_bn_act(stream(A.device), DT[A.dtype], ptr(A), M, K, ptr(W), N, ptr(scale), ptr(shift), act,
                                   ptr(residual), ptr(gate), rows_per_image, ptr(C), 0, 0)
    assert rc == 0, rc
    return C


def sepconv(dtype, B, level_hw, level_inputs, fuse_mode, fw, den, pre_act, taps, wq, scale, shift, affine_rows, post_act,
            F, N, outs, out_strides, ood=None, A=9):
    """level_inputs[l] = [(tensor_ptr, image_stride, (H, W), mode), ...]"""
    lib = _lib.load()
    arr = lambda ct, v: (ct * len(v))(*v)
    nl, n_in = len(level_hw), len(level_inputs[0])
    c_hw = arr(ctypes.c_int, [v for hw in level_hw for v in hw])
    c_ptr = arr(ctypes.c_void_p, [i[0] for lv in level_inputs for i in lv])
    c_str = arr(ctypes.c_longlong, [i[1] for lv in level_inputs for i in lv])
    c_ihw = arr(ctypes.c_int, [v for lv in level_inputs for i in lv for v in i[2]])
    c_mode = arr(ctypes.c_int, [i[3] for lv in level_inputs for i in lv])
    c_fw = arr(ctypes.c_float, list(fw) + [0.0] * (3 - len(fw)))
    c_aff = arr(ctypes.c_int, affine_rows)
    c_out = arr(ctypes.c_void_p, outs)
    c_ostr = arr(ctypes.c_longlong, out_strides)
    if ood is None:
        oa = (0, A, None, None, 0, None)
    else:
        oa = (ood['classes'], A, ptr(ood['energy']), ptr(ood['maxlogit']), ood['stride'], arr(ctypes.c_longlong, ood['level_off']))
    dev = taps.device
    rc = lib.effdet_sepconv_fused(stream(dev), dtype if isinstance(dtype, int) else DT[dtype], B, nl, c_hw, n_in, c_ptr, c_str, c_ihw, c_mode, fuse_mode, c_fw,
                                  den, pre_act, ptr(taps), ptr(wq), ptr(scale), ptr(shift), c_aff, post_act, F, N,
                                  c_out, c_ostr, *oa)
    assert rc == 0, rc
