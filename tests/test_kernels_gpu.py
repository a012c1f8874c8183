"""Kernel-level parity: every C-ABI entry point against the CPU oracle's arithmetic (-m gpu)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import model as om
from oracle import postprocess as op

DEV = 'cuda:0'
TOL = {torch.float32: 2e-5, torch.bfloat16: 3e-2}     # relative to max|ref|


def _rel(a, b):
    return float((a.float().cpu() - b.float().cpu()).abs().max() / (b.float().abs().max() + 1e-12))


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('M,K,N', [(300, 16, 96), (1000, 96, 24), (257, 240, 40), (513, 1152, 320), (128, 40, 240), (77, 64, 810), (300, 8, 16)])
def test_pw_gemm(dtype, M, K, N):
    import _hip
    A, W = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5)
    scale, shift = torch.rand(N) + 0.5, _rand(N, seed=3, scale=0.1)
    res = _rand(M, N, seed=4)
    Aq, Wq, resq = A.to(dtype), W.to(dtype), res.to(dtype)
    for act, use_res in ((0, False), (1, False), (0, True)):
        ref = (Aq.double() @ Wq.double().t()) * scale.double() + shift.double()
        if act:
            ref = ref * torch.sigmoid(ref)
        if use_res:
            ref = ref + resq.double()
        out = _hip.pw_gemm(Aq.to(DEV), Wq.to(DEV), scale.to(DEV), shift.to(DEV), act, resq.to(DEV) if use_res else None)
        assert _rel(out, ref) < TOL[dtype]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_pw_gemm_group_equals_single_launches(dtype):
    """effdet_pw_gemm_group (the BiFPN's lateral convs in one launch) is bit-identical to one effdet_pw_gemm_bn_act per problem"""
    import ctypes
    import _hip
    from ood_object_detection_amd import _lib
    lib = _lib.load()
    st = _hip.stream(DEV)
    N = 64
    probs = [(2 * 400, 320), (2 * 1600, 112), (2 * 6400, 40), (2 * 100, 320), (77, 8)]
    keep, single, grouped = [], [], []
    for i, (M, K) in enumerate(probs):
        A = _rand(M, K, seed=30 + i).to(dtype).to(DEV)
        W = _rand(N, K, seed=40 + i, scale=K ** -0.5).to(dtype).to(DEV)
        sc = (torch.rand(N) + 0.5).to(DEV) if i % 2 == 0 else None
        sh = _rand(N, seed=50 + i, scale=0.1).to(DEV)
        c0, c1 = torch.empty(M, N, dtype=dtype, device=DEV), torch.empty(M, N, dtype=dtype, device=DEV)
        assert lib.effdet_pw_gemm_bn_act(st, _hip.DT[dtype], A.data_ptr(), M, K, W.data_ptr(), N, None if sc is None else sc.data_ptr(),
                                         sh.data_ptr(), 0, None, None, 0, c0.data_ptr(), 0, 0) == 0
        keep.append((A, W, sc, sh))
        single.append(c0)
        grouped.append(c1)
    n = len(probs)
    arr = lambda ct, v: (ct * n)(*v)
    rc = lib.effdet_pw_gemm_group(st, _hip.DT[dtype], n, arr(ctypes.c_void_p, [k[0].data_ptr() for k in keep]),
                                  arr(ctypes.c_longlong, [p[0] for p in probs]), arr(ctypes.c_int, [p[1] for p in probs]),
                                  arr(ctypes.c_void_p, [k[1].data_ptr() for k in keep]), arr(ctypes.c_int, [N] * n),
                                  arr(ctypes.c_void_p, [None if k[2] is None else k[2].data_ptr() for k in keep]),
                                  arr(ctypes.c_void_p, [k[3].data_ptr() for k in keep]), 0,
                                  arr(ctypes.c_void_p, [c.data_ptr() for c in grouped]))
    assert rc == 0
    torch.cuda.synchronize()
    for a, b in zip(single, grouped):
        assert torch.equal(a, b)
    # mixed output tiles are refused
    assert lib.effdet_pw_gemm_group(st, _hip.DT[dtype], 2, arr(ctypes.c_void_p, [keep[0][0].data_ptr()] * n), arr(ctypes.c_longlong, [800] * n),
                                    arr(ctypes.c_int, [320] * n), arr(ctypes.c_void_p, [keep[0][1].data_ptr()] * n), arr(ctypes.c_int, [64, 24, 0, 0, 0]),
                                    arr(ctypes.c_void_p, [None] * n), arr(ctypes.c_void_p, [keep[0][3].data_ptr()] * n), 0,
                                    arr(ctypes.c_void_p, [grouped[0].data_ptr()] * n)) != 0


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_pw_gemm_gate_and_strided_out(dtype):
    import _hip
    from ood_object_detection_amd import _lib
    B, HW, K, N = 3, 50, 48, 24
    A, W = _rand(B * HW, K, seed=5).to(dtype), _rand(N, K, seed=6, scale=K ** -0.5).to(dtype)
    gate = torch.rand(B, K)
    shift = _rand(N, seed=7, scale=0.1)
    Ag = (A.float().reshape(B, HW, K) * gate[:, None, :]).to(dtype).reshape(B * HW, K)
    ref = Ag.double() @ W.double().t() + shift.double()
    out = _hip.pw_gemm(A.to(DEV), W.to(DEV), None, shift.to(DEV), 0, None, gate.to(DEV), HW)
    assert _rel(out, ref) < TOL[dtype]
    # strided output: rows of image b land at b*stride + p*ldc
    lib = _lib.load()
    ldc, istride = N + 8, HW * (N + 8) + 16
    C = torch.zeros(B * istride, dtype=dtype, device=DEV)
    Ad, Wd, sd = A.to(DEV), W.to(DEV), shift.to(DEV)
    rc = lib.effdet_pw_gemm_bn_act(_hip.stream(DEV), _hip.DT[dtype], Ad.data_ptr(), B * HW, K, Wd.data_ptr(), N, None,
                                   sd.data_ptr(), 0, None, None, HW, C.data_ptr(), istride, ldc)
    assert rc == 0
    ref2 = (A.double() @ W.double().t() + shift.double()).reshape(B, HW, N)
    got = torch.stack([C[b * istride:b * istride + HW * ldc].reshape(HW, ldc)[:, :N] for b in range(B)])
    assert _rel(got, ref2) < TOL[dtype]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('C,H,W,k,s', [(32, 20, 20, 3, 1), (96, 22, 18, 3, 2), (144, 17, 17, 5, 2), (240, 9, 12, 5, 1), (1152, 5, 5, 3, 1)])
def test_dwconv_se(dtype, C, H, W, k, s):
    import _hip
    from ood_object_detection_amd import _lib
    lib = _lib.load()
    B, R = 2, max(1, C // 24)
    x = _rand(B, C, H, W, seed=8).to(dtype)
    w = _rand(C, 1, k, k, seed=9, scale=1.0 / k)
    scale, shift = torch.rand(C) + 0.5, _rand(C, seed=10, scale=0.1)
    ref = om.conv2d_pad(x.float(), w, None, s, 'same', groups=C) * scale[None, :, None, None] + shift[None, :, None, None]
    ref = om.silu(ref)
    Ho, Wo = ref.shape[2], ref.shape[3]
    xd = _hip.nhwc(x, dtype).to(DEV)
    y = torch.empty(B, Ho, Wo, C, dtype=dtype, device=DEV)
    taps = w.permute(2, 3, 0, 1).reshape(k * k, C).contiguous().to(DEV)
    nblk = lib.effdet_dwconv_blocks_per_image(Ho, Wo, C)
    assert nblk > 0
    part = torch.zeros(B, nblk, C, dtype=torch.float32, device=DEV)
    sd, td = scale.to(DEV), shift.to(DEV)
    rc = lib.effdet_dwconv_bn_act(_hip.stream(DEV), _hip.DT[dtype], xd.data_ptr(), y.data_ptr(), taps.data_ptr(), sd.data_ptr(),
                                  td.data_ptr(), 1, part.data_ptr(), B, H, W, C, k, s)
    assert rc == 0
    assert _rel(_hip.nchw(y), ref) < TOL[dtype]
    # SE gate from the partial sums
    W1, b1 = _rand(R, C, seed=11, scale=C ** -0.5), _rand(R, seed=12, scale=0.1)
    W2, b2 = _rand(C, R, seed=13, scale=R ** -0.5), _rand(C, seed=14, scale=0.1)
    pooled = _hip.nchw(y).cpu().mean((2, 3))
    gref = torch.sigmoid(om.silu(pooled @ W1.t() + b1) @ W2.t() + b2)
    gate = torch.empty(B, C, dtype=torch.float32, device=DEV)
    args = [t.to(DEV).contiguous() for t in (W1, b1, W2.t(), b2)]       # conv_expand weight goes in transposed ([R][C])
    rc = lib.effdet_se_gate(_hip.stream(DEV), part.data_ptr(), nblk, Ho * Wo, *[a.data_ptr() for a in args], gate.data_ptr(), B, C, R)
    assert rc == 0
    assert _rel(gate, gref) < 2e-5


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('Cin,mid,H,W,k,s', [(16, 96, 40, 36, 3, 2), (24, 144, 22, 30, 3, 1), (24, 144, 33, 21, 5, 2),
                                             (40, 240, 20, 20, 5, 1), (112, 672, 10, 12, 5, 2), (192, 1152, 5, 5, 3, 1),
                                             (80, 480, 40, 40, 3, 1), (112, 672, 40, 40, 5, 2), (192, 1152, 20, 20, 5, 1), (112, 680, 23, 17, 5, 1),
                                             # wide inputs on the shared-X rolling-window form (mbconv_wide.hip): every d0 late-stage shape, odd
                                             # sizes, two column strips, several bands
                                             (80, 480, 40, 40, 5, 1), (112, 672, 40, 40, 5, 1), (192, 1152, 20, 20, 3, 1), (112, 672, 37, 41, 5, 1),
                                             (80, 480, 33, 40, 3, 2), (80, 480, 24, 64, 3, 1), (192, 1152, 13, 19, 5, 2), (160, 960, 32, 40, 5, 1), (136, 816, 24, 24, 3, 1), (160, 960, 33, 31, 5, 2),
                                             # two channel tiles per wave (stride-2 3 x 3 blocks with >= 32 input channels, mbconv_roll.hip NJ = 2)
                                             (32, 96, 37, 45, 3, 2), (32, 192, 30, 70, 3, 2)])
def test_mbconv_expand_dw_fused(dtype, Cin, mid, H, W, k, s):
    """fused expand 1x1 + BN + SiLU -> depthwise + BN + SiLU + SE pool partials vs the oracle's separate ops"""
    import _hip
    from ood_object_detection_amd import _lib
    lib = _lib.load()
    B = 2
    x = _rand(B, Cin, H, W, seed=50).to(dtype)
    w1 = _rand(mid, Cin, 1, 1, seed=51, scale=1.5 * Cin ** -0.5).to(dtype)
    s1, t1 = torch.rand(mid) + 0.5, _rand(mid, seed=52, scale=0.2)
    wd = _rand(mid, 1, k, k, seed=53, scale=1.0 / k)
    s2, t2 = torch.rand(mid) + 0.5, _rand(mid, seed=54, scale=0.2)
    e = om.silu(F.conv2d(x.float(), w1.float()) * s1[None, :, None, None] + t1[None, :, None, None])
    if dtype == torch.bfloat16:
        e = e.to(dtype).float()                      # the kernel keeps the expanded map in LDS as bf16
    ref = om.silu(om.conv2d_pad(e, wd, None, s, 'same', groups=mid) * s2[None, :, None, None] + t2[None, :, None, None])
    Ho, Wo = ref.shape[2], ref.shape[3]
    xd = _hip.nhwc(x, dtype).to(DEV)
    y = torch.empty(B, Ho, Wo, mid, dtype=dtype, device=DEV)
    nt = lib.effdet_mbconv_tiles_per_image(_hip.DT[dtype], H, W, Cin, mid, k, s)
    assert nt > 0
    part = torch.full((B, nt, mid), float('nan'), dtype=torch.float32, device=DEV)
    dv = [t.contiguous().to(DEV) for t in (w1.reshape(mid, Cin), s1, t1, wd.permute(2, 3, 0, 1).reshape(k * k, mid), s2, t2)]
    rc = lib.effdet_mbconv_expand_dw(_hip.stream(DEV), _hip.DT[dtype], xd.data_ptr(), y.data_ptr(), *[t.data_ptr() for t in dv],
                                     part.data_ptr(), B, H, W, Cin, mid, k, s)
    assert rc == 0
    assert _rel(_hip.nchw(y), ref) < TOL[dtype]
    _check_pool_against_oracle(part, Ho, Wo, ref, e, wd, s2, k, s, dtype)


def test_wide_handoff_timeout_is_reported():
    """mbconv_wide.hip hands X rows over through LDS arrival counters with a BOUNDED spin.  A wave that runs out of spins goes on
    (no hung grid) but must say so: it sets the library's device-side failure word, and the next launching effdet_* call returns
    -5 with a text that names it.  Seen once through the test-only variant build whose spin limit is 0
    (_lib.TEST_VARIANTS['spin0'], built by __graft_entry__.build(); the package never loads it); the product library run on
    the same problem leaves the word clear."""
    import ctypes
    import os
    import _hip
    from ood_object_detection_amd import _lib
    lib = _lib.load()
    vpath = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libeffdet_hip_spin0.so')
    assert os.path.exists(vpath), 'run __graft_entry__.build() (it builds the test variants)'
    var = ctypes.CDLL(vpath)
    B, Cin, mid, H, W, k, s = 2, 80, 480, 40, 40, 3, 1
    assert lib.effdet_mbconv_tiles_per_image(1, H, W, Cin, mid, k, s) > 0
    x = _rand(B, Cin, H, W, seed=50).to(torch.bfloat16)
    xd = _hip.nhwc(x, torch.bfloat16).to(DEV)
    y = torch.empty(B, H, W, mid, dtype=torch.bfloat16, device=DEV)
    dv = [t.contiguous().to(DEV) for t in (_rand(mid, Cin, seed=51).to(torch.bfloat16), torch.ones(mid), torch.zeros(mid),
                                           _rand(k * k, mid, seed=53), torch.ones(mid), torch.zeros(mid))]

    def run(l):
        fn = l.effdet_mbconv_expand_dw
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 9 + [ctypes.c_int] * 7
        return fn(_hip.stream(DEV), 1, xd.data_ptr(), y.data_ptr(), *[t.data_ptr() for t in dv], None, B, H, W, Cin, mid, k, s)

    assert run(lib) == 0
    torch.cuda.synchronize()
    assert lib.effdet_device_error(0) == 0 and run(lib) == 0          # the real hand-off never times out
    torch.cuda.synchronize()
    var.effdet_device_error.restype = ctypes.c_int
    var.effdet_device_error.argtypes = [ctypes.c_int]
    var.effdet_last_error.restype = ctypes.c_char_p
    assert var.effdet_device_error(1) == 0
    assert run(var) == 0                                              # the launch itself succeeds ...
    torch.cuda.synchronize()
    assert var.effdet_device_error(0) & 1                             # ... its waves gave up and said so
    assert run(var) == -5                                             # the next call reports it
    assert b'device-side failure' in var.effdet_last_error()
    torch.cuda.synchronize()
    assert var.effdet_device_error(1) & 1 and var.effdet_device_error(0) == 0     # cleared
    assert lib.effdet_device_error(0) == 0                            # the product library's own word was never touched


def _check_pool_against_oracle(part, Ho, Wo, ref, e, wd, s2, k, stride, dtype):
    """SE pool partial sums against the ORACLE's pooled activation (not against the kernel's own output).  float32: 1e-4.
    bf16: the bound follows from the roundings the kernel makes, it is not a measured band: the expanded map is stored as bf16
    (relative 2^-9 per element, as is the bf16 weight fold of BN1's scale) and travels through the depthwise taps, BN2's scale
    and SiLU (slope <= 1.1), so an output element is off by at most 1.1 * |s2| * sum_taps |w| |e| * 2^-9 for each of the two;
    the rolling-window kernels pool the unrounded float32 value, the tile forms the value after the final rounding (one more
    2^-9 of |y|).  Pooling averages these bounds (signs are not assumed to cancel)."""
    pooled = part.sum(1).cpu() / (Ho * Wo)
    pref = ref.mean((2, 3))
    if dtype == torch.float32:
        assert float((pooled - pref).abs().max()) < 1e-4 * max(1.0, float(pref.abs().max()))
        return
    amp = om.conv2d_pad(e.abs(), wd.abs(), None, stride, 'same', groups=wd.shape[0]) * s2.abs()[None, :, None, None]
    bound = 2.0 ** -9 * (2 * 1.1 * amp + ref.abs()).mean((2, 3)) + 1e-5
    worst = float(((pooled - pref).abs() / bound).max())
    assert worst < 1.0, worst


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('Cin,mid,H,W,k,s', [(32, 96, 40, 36, 3, 2), (32, 96, 64, 64, 3, 2), (16, 48, 21, 50, 5, 1)])
def test_mbconv_expand_dw_gated(dtype, Cin, mid, H, W, k, s):
    """the gated form (block 1.0 of the backbone: block 0.0's project conv is composed into the expand weights, its SE gate
    multiplies the input along K): expand(x * gate) -> depthwise, vs the oracle's separate ops"""
    import _hip
    from ood_object_detection_amd import _lib
    lib = _lib.load()
    B = 3
    x = _rand(B, Cin, H, W, seed=70).to(dtype)
    gate = torch.sigmoid(_rand(B, Cin, seed=75))
    w1 = _rand(mid, Cin, 1, 1, seed=71, scale=1.5 * Cin ** -0.5).to(dtype)
    s1, t1 = torch.rand(mid) + 0.5, _rand(mid, seed=72, scale=0.2)
    wd = _rand(mid, 1, k, k, seed=73, scale=1.0 / k)
    s2, t2 = torch.rand(mid) + 0.5, _rand(mid, seed=74, scale=0.2)
    xg = (x.float() * gate[:, :, None, None]).to(dtype).float()
    e = om.silu(F.conv2d(xg, w1.float()) * s1[None, :, None, None] + t1[None, :, None, None])
    if dtype == torch.bfloat16:
        e = e.to(dtype).float()
    ref = om.silu(om.conv2d_pad(e, wd, None, s, 'same', groups=mid) * s2[None, :, None, None] + t2[None, :, None, None])
    Ho, Wo = ref.shape[2], ref.shape[3]
    xd = _hip.nhwc(x, dtype).to(DEV)
    y = torch.empty(B, Ho, Wo, mid, dtype=dtype, device=DEV)
    nt = lib.effdet_mbconv_gated_tiles_per_image(_hip.DT[dtype], H, W, Cin, mid, k, s)
    assert nt > 0
    part = torch.full((B, nt, mid), float('nan'), dtype=torch.float32, device=DEV)
    gd = gate.contiguous().to(DEV)
    dv = [t.contiguous().to(DEV) for t in (w1.reshape(mid, Cin), s1, t1, wd.permute(2, 3, 0, 1).reshape(k * k, mid), s2, t2)]
    rc = lib.effdet_mbconv_expand_dw_gated(_hip.stream(DEV), _hip.DT[dtype], xd.data_ptr(), gd.data_ptr(), y.data_ptr(),
                                           *[t.data_ptr() for t in dv], part.data_ptr(), B, H, W, Cin, mid, k, s)
    assert rc == 0
    assert _rel(_hip.nchw(y), ref) < TOL[dtype]
    _check_pool_against_oracle(part, Ho, Wo, ref, e, wd, s2, k, s, dtype)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('C,H,W', [(32, 64, 96), (48, 70, 38), (40, 34, 34), (32, 33, 35), (32, 100, 172), (32, 37, 64)])   # (32, 33, 35): odd width -> scalar patch load
def test_stem_dw_fused(dtype, C, H, W):
    """conv_stem + BN + SiLU -> depthwise 3x3 + BN + SiLU (+ pool partials) in one launch vs the oracle ops"""
    import _hip
    from ood_object_detection_amd import _lib
    lib = _lib.load()
    B = 2
    x = _rand(B, 3, H, W, seed=60).to(dtype)
    w = _rand(C, 3, 3, 3, seed=61, scale=0.25).to(dtype)
    s1, t1 = torch.rand(C) + 0.5, _rand(C, seed=62, scale=0.2)
    wd = _rand(C, 1, 3, 3, seed=63, scale=0.35)
    s2, t2 = torch.rand(C) + 0.5, _rand(C, seed=64, scale=0.2)
    e = om.silu(om.conv2d_pad(x.float(), w.float(), None, 2, 'same') * s1[None, :, None, None] + t1[None, :, None, None])
    if dtype == torch.bfloat16:
        e = e.to(dtype).float()
    ref = om.silu(om.conv2d_pad(e, wd, None, 1, 'same', groups=C) * s2[None, :, None, None] + t2[None, :, None, None])
    Ho, Wo = ref.shape[2], ref.shape[3]
    wk = torch.zeros(C, 32)
    wk[:, :27] = w.float().permute(0, 2, 3, 1).reshape(C, 27)
    nt = lib.effdet_stem_dw_parts(_hip.DT[dtype], H, W, C)       # bf16 with C = 32 and an even left pad: the rolling-window form
    y = torch.empty(B, Ho, Wo, C, dtype=dtype, device=DEV)
    part = torch.full((B, nt, C), float('nan'), dtype=torch.float32, device=DEV)
    dv = [t.contiguous().to(DEV) for t in (wk.to(dtype), s1, t1, wd.permute(2, 3, 0, 1).reshape(9, C), s2, t2)]
    xd = x.to(DEV)
    rc = lib.effdet_stem_dw_fused(_hip.stream(DEV), _hip.DT[dtype], _hip.DT[dtype], xd.data_ptr(), *[t.data_ptr() for t in dv],
                                  y.data_ptr(), part.data_ptr(), B, H, W, C)
    assert rc == 0
    assert _rel(_hip.nchw(y), ref) < TOL[dtype]
    _check_pool_against_oracle(part, Ho, Wo, ref, e, wd, s2, 3, 1, dtype)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_stem_and_maxpool(dtype):
    import _hip
    from ood_object_detection_amd import _lib
    lib = _lib.load()
    B, H, W, Co = 2, 64, 96, 32
    x = _rand(B, 3, H, W, seed=15)
    w = _rand(Co, 3, 3, 3, seed=16, scale=0.2)
    scale, shift = torch.rand(Co) + 0.5, _rand(Co, seed=17, scale=0.1)
    ref = om.silu(om.conv2d_pad(x, w, None, 2, 'same') * scale[None, :, None, None] + shift[None, :, None, None])
    xd = x.to(DEV)
    y = torch.empty(B, H // 2, W // 2, Co, dtype=dtype, device=DEV)
    wt = w.permute(2, 3, 1, 0).reshape(27, Co).contiguous().to(DEV)
    sd, td = scale.to(DEV), shift.to(DEV)
    rc = lib.effdet_stem_conv(_hip.stream(DEV), 0, _hip.DT[dtype], xd.data_ptr(), wt.data_ptr(), sd.data_ptr(), td.data_ptr(),
                              y.data_ptr(), B, H, W, Co)
    assert rc == 0
    assert _rel(_hip.nchw(y), ref) < TOL[dtype]
    for (h, w_) in ((20, 20), (10, 10), (5, 7)):
        f = _rand(B, 64, h, w_, seed=18).to(dtype)
        pref = om.maxpool_pad(f.float(), 3, 2, 'same')
        fd = _hip.nhwc(f, dtype).to(DEV)
        out = torch.empty(B, pref.shape[2], pref.shape[3], 64, dtype=dtype, device=DEV)
        rc = lib.effdet_maxpool_same(_hip.stream(DEV), _hip.DT[dtype], fd.data_ptr(), 0, out.data_ptr(), 0, B, h, w_, 64)
        assert rc == 0
        assert torch.equal(_hip.nchw(out).cpu(), pref)


def _sep_ref(ins, modes, fw, den, fuse_mode, pre_act, dw, pw, bias, scale, shift, post_act):
    """oracle arithmetic of one fused node: combine -> act -> dw3x3 -> pw -> affine -> act"""
    xs = []
    for x, m in zip(ins, modes):
        if m == 1:
            x = F.interpolate(x, scale_factor=2.0, mode='nearest')
        elif m == 2:
            x = om.maxpool_pad(x, 3, 2, 'same')
        xs.append(x)
    if fuse_mode == 0:
        y = xs[0]
    elif fuse_mode == 1:
        y = sum((x * w) / den for x, w in zip(xs, fw))
    else:
        y = sum(x * w for x, w in zip(xs, fw))
    if pre_act:
        y = om.silu(y)
    y = om.conv2d_pad(y, dw, None, 1, 'same', groups=y.shape[1])
    y = F.conv2d(y, pw, bias)
    if scale is not None:
        y = y * scale[None, :, None, None]
    y = y + shift[None, :, None, None]
    return om.silu(y) if post_act else y


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('Fc', [64, 88, 112])
def test_sepconv_bifpn_node(dtype, Fc):
    import _hip
    B, H, W = 2, 20, 12
    x_same = _rand(B, Fc, H, W, seed=20).to(dtype)
    x_up = _rand(B, Fc, H // 2, W // 2, seed=21).to(dtype)
    x_dn = _rand(B, Fc, 2 * H, 2 * W - 1, seed=22).to(dtype)      # odd width: SAME pad on one side only
    dw = _rand(Fc, 1, 3, 3, seed=23, scale=0.3)
    pw = _rand(Fc, Fc, 1, 1, seed=24, scale=Fc ** -0.5).to(dtype)
    scale, shift = torch.rand(Fc) + 0.5, _rand(Fc, seed=25, scale=0.1)
    fw, den = [0.7, 1.3, 0.4], 2.4001
    ref = _sep_ref([x_same.float(), x_up.float(), x_dn.float()], [0, 1, 2], fw, den, 1, 1, dw, pw.float(), None, scale, shift, 0)
    ins_d = [_hip.nhwc(x, dtype).to(DEV) for x in (x_same, x_up, x_dn)]
    out = torch.empty(B, H, W, Fc, dtype=dtype, device=DEV)
    taps = dw.permute(2, 3, 0, 1).reshape(9, Fc).contiguous().to(DEV)
    wq = pw.reshape(Fc, Fc).contiguous().to(DEV)
    sd, td = scale.to(DEV), shift.to(DEV)
    li = [[(t.data_ptr(), t.shape[1] * t.shape[2] * t.shape[3], (t.shape[1], t.shape[2]), m) for t, m in zip(ins_d, (0, 1, 2))]]
    _hip.sepconv(dtype, B, [(H, W)], li, 1, fw, den, 1, taps, wq, sd, td, [0], 0, Fc, Fc, [out.data_ptr()], [H * W * Fc])
    assert _rel(_hip.nchw(out), ref) < TOL[dtype]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('C', [90, 7, 1, 150, 2, 20, 32, 64, 96])
def test_sepconv_head_levels_and_ood(dtype, C):
    """all pyramid levels in one launch, per-level affine, class-predict layout + OOD epilogue"""
    import _hip
    B, Fc, A = 2, 64, 9
    hw = [(16, 16), (8, 8), (4, 4), (2, 2), (1, 1)]
    offs = np.cumsum([0] + [h * w for h, w in hw]).tolist()
    P = offs[-1]
    feats = [_rand(B, Fc, h, w, seed=30 + i).to(dtype) for i, (h, w) in enumerate(hw)]
    pyr = torch.cat([_hip.nhwc(f, dtype).reshape(B, -1, Fc) for f in feats], 1).contiguous().to(DEV)
    dw = _rand(Fc, 1, 3, 3, seed=36, scale=0.3)
    taps = dw.permute(2, 3, 0, 1).reshape(9, Fc).contiguous().to(DEV)
    es = pyr.element_size()
    # (a) tower layer: per-level BN + SiLU
    pw = _rand(Fc, Fc, 1, 1, seed=37, scale=Fc ** -0.5).to(dtype)
    scale, shift = torch.rand(5, Fc) + 0.5, _rand(5, Fc, seed=38, scale=0.1)
    out = torch.empty(B, P, Fc, dtype=dtype, device=DEV)
    li = [[(pyr.data_ptr() + offs[l] * Fc * es, P * Fc, hw[l], 0)] for l in range(5)]
    _hip.sepconv(dtype, B, hw, li, 0, [], 1.0, 0, taps, pw.reshape(Fc, Fc).contiguous().to(DEV), scale.to(DEV), shift.to(DEV),
                 list(range(5)), 1, Fc, Fc, [out.data_ptr() + offs[l] * Fc * es for l in range(5)], [P * Fc] * 5)
    for l in range(5):
        ref = _sep_ref([feats[l].float()], [0], [], 1.0, 0, 0, dw, pw.float(), None, scale[l], shift[l], 1)
        got = out[:, offs[l]:offs[l + 1], :].reshape(B, hw[l][0], hw[l][1], Fc)
        assert _rel(_hip.nchw(got), ref) < TOL[dtype]
    # (b) class predict with OOD epilogue
    NO = A * C
    pwp = _rand(NO, Fc, 1, 1, seed=39, scale=2.0 * Fc ** -0.5).to(dtype)
    bias = _rand(NO, seed=40, scale=0.5) - 2.0
    N = A * P
    cls_all = torch.empty(B, N, C, dtype=dtype, device=DEV)
    energy = torch.empty(B, N, dtype=torch.float32, device=DEV)
    maxl = torch.empty(B, N, dtype=torch.float32, device=DEV)
    _hip.sepconv(dtype, B, hw, li, 0, [], 1.0, 0, taps, pwp.reshape(NO, Fc).contiguous().to(DEV), None, bias.reshape(1, NO).to(DEV),
                 [0] * 5, 0, Fc, NO, [cls_all.data_ptr() + offs[l] * NO * es for l in range(5)], [P * NO] * 5,
                 ood=dict(classes=C, energy=energy, maxlogit=maxl, stride=N, level_off=[o * A for o in offs[:5]]), A=A)
    refs = [_sep_ref([feats[l].float()], [0], [], 1.0, 0, 0, dw, pwp.float(), bias, None, torch.zeros(NO), 0) for l in range(5)]
    ref_all = torch.cat([r.permute(0, 2, 3, 1).reshape(B, -1, C) for r in refs], 1)
    assert _rel(cls_all, ref_all) < TOL[dtype]
    e_ref, m_ref = om.ood_scores(refs, C)
    tol = 1e-4 if dtype == torch.float32 else 5e-2
    assert float((energy.cpu() - e_ref).abs().max()) < tol * max(1.0, float(e_ref.abs().max()))
    assert float((maxl.cpu() - m_ref).abs().max()) < tol * max(1.0, float(m_ref.abs().max()))


# ------------------------------------------------------------------------------------ post-process
def _pp_case(seed, B, C, sizes, A=9, cs=2.0, shift=0.0, bs=0.4):
    from _seeded import seeded_array
    cls = [torch.from_numpy(seeded_array(seed, 'cls%d' % i, (B, A * C, s, s), scale=cs)) - shift for i, s in enumerate(sizes)]
    box = [torch.from_numpy(seeded_array(seed, 'box%d' % i, (B, A * 4, s, s), scale=bs)) for i, s in enumerate(sizes)]
    return cls, box


def _check_topk(cls, box, C, k, dtype=torch.float32):
    from ood_object_detection_amd.effdet.bench import _post_process
    cls_q = [c.to(dtype) for c in cls]
    box_q = [b.to(dtype) for b in box]
    rc, rb, ri, rcl = op.post_process([c.float() for c in cls_q], [b.float() for b in box_q], len(cls), C, k)
    B = cls[0].shape[0]
    # per-anchor maxima of the UNROUNDED logits, as the class head emits them: rounding is monotonic, so the
    # select must round them itself to agree with the stored logits
    amax = torch.cat([c.permute(0, 2, 3, 1).reshape(B, -1, C) for c in cls], 1).float().max(dim=2).values
    for am in (None, amax.to(DEV)):
        gc, gb, gi, gcl = _post_process([c.to(DEV) for c in cls_q], [b.to(DEV) for b in box_q], len(cls), C, k, anchor_max=am)
        assert torch.equal(gi.cpu(), ri) and torch.equal(gcl.cpu(), rcl)
        assert torch.equal(gc.float().cpu(), rc) and torch.equal(gb.float().cpu(), rb)


def test_topk_golden(golden):
    g = golden('post_process')
    B, C, A, k = [int(v) for v in g['meta'][:4]]
    sizes = [int(v) for v in g['meta'][4:]]
    cls, box = _pp_case(1, B, C, sizes, A, 2.0, bs=0.5)
    from ood_object_detection_amd.effdet.bench import _post_process
    gc, gb, gi, gcl = _post_process([c.to(DEV) for c in cls], [b.to(DEV) for b in box], 5, C, k)
    assert np.array_equal(gi.cpu().numpy(), g['indices']) and np.array_equal(gcl.cpu().numpy(), g['classes'])
    assert np.array_equal(gc.cpu().numpy(), g['cls_topk']) and np.array_equal(gb.cpu().numpy(), g['box_topk'])


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_topk_vs_oracle_with_ties(dtype):
    cls, box = _pp_case(7, 3, 11, [16, 8, 4, 2, 1])
    _check_topk(cls, box, 11, 1000, dtype)          # bf16: thousands of exact ties -> index order must hold
    _check_topk(cls, box, 11, 500, dtype)           # 3069 anchors >= 4k: the anchor-prefilter path


def test_topk_degenerate_all_equal_and_large():
    # every logit identical: selection is decided purely by index bits (all six radix passes run)
    B, C, A = 2, 5, 9
    sizes = [32, 16, 8, 4, 2]
    cls = [torch.full((B, A * C, s, s), -4.59512) for s in sizes]
    box = [torch.zeros(B, A * 4, s, s) for s in sizes]
    _check_topk(cls, box, C, 5000)
    _check_topk(cls, box, C, 1000)                  # prefilter with every anchor tied: the anchor set is all of them
    # heavy-tie block bigger than the LDS sort buffer (forces a second radix pass) + k == CAP boundary
    cls2, box2 = _pp_case(9, 1, 90, [40, 20, 10, 5, 3], cs=0.01, shift=4.6)
    _check_topk(cls2, box2, 90, 5000)
    _check_topk(cls2, box2, 90, 4000)               # same through the anchor prefilter (19206 anchors >= 4k)


@pytest.mark.parametrize('soft', [False, True])
def test_generate_detections_golden(golden, soft):
    from ood_object_detection_amd.effdet.anchors import generate_detections
    g = golden('generate_detections')
    B = int(g['meta'][0])
    anchors = torch.from_numpy(g['anchors']).to(DEV)
    tag = 'soft' if soft else 'hard'
    for i in range(B):
        args = [torch.from_numpy(g[k][i]).to(DEV) for k in ('cls_topk', 'box_topk')] + [anchors] + \
               [torch.from_numpy(g[k][i]).to(DEV) for k in ('indices', 'classes')]
        for key, extra in (('det_%s_%d', (None, torch.tensor(128), 100)),
                           ('det_%s_info_%d', (torch.from_numpy(g['img_scale'])[i], torch.from_numpy(g['img_size'])[i], 20))):
            det = generate_detections(*args, extra[0], extra[1], max_det_per_image=extra[2], soft_nms=soft).cpu().numpy()
            ref = g[key % (tag, i)]
            assert det.shape == ref.shape
            assert np.array_equal(det[:, 5], ref[:, 5])                       # classes exact
            assert np.abs(det[:, :4] - ref[:, :4]).max() <= 1e-4              # pixels (exp / sigmoid ulps)
            assert np.abs(det[:, 4] - ref[:, 4]).max() <= 1e-6


@pytest.mark.parametrize('gather', [False, True])
def test_detections_hard_one_launch_equals_decode_then_nms(gather):
    """effdet_detections_hard (decode + threshold + hard NMS in one launch, what batched_detections runs) is bit-identical to
    effdet_decode_threshold[_gather] followed by effdet_nms_hard, intermediate arrays included"""
    import _hip
    from ood_object_detection_amd import _lib
    lib = _lib.load()
    B, k, N = 3, 700, 4000
    g = torch.Generator().manual_seed(9)
    anchors = torch.rand(N, 4, generator=g) * 200
    anchors[:, 2:] += anchors[:, :2] + 8
    idx = torch.stack([torch.randperm(N, generator=g)[:k] for _ in range(B)]).to(torch.int64)
    cls_id = torch.randint(0, 7, (B, k), generator=g)
    logit = torch.randn(B, k, generator=g) * 2 - 1
    box_all = torch.randn(B, N, 4, generator=g) * 0.3
    box_topk = torch.gather(box_all, 1, idx[:, :, None].expand(B, k, 4)).contiguous()
    scale, size = torch.tensor([1.0, 1.5, 0.8]), torch.tensor([[300., 280.], [256., 256.], [320., 200.]])
    dev = lambda t: t.contiguous().to(DEV)
    a_d, i_d, c_d, l_d, sc_d, sz_d = dev(anchors), dev(idx), dev(cls_id), dev(logit), dev(scale), dev(size)
    bsrc = dev(box_all) if gather else dev(box_topk)

    def bufs():
        f = dict(device=DEV, dtype=torch.float32); i = dict(device=DEV, dtype=torch.int32)
        return [torch.zeros(B, k, 4, **f), torch.zeros(B, k, **f), torch.zeros(B, k, **i), torch.zeros(B, k, **i), torch.zeros(B, **i), torch.zeros(B, **f),
                torch.zeros(B, 100, 6, **f), torch.zeros(B, **i), torch.zeros(B, 100, **i)]
    st = _hip.stream(DEV)
    one, two = bufs(), bufs()
    tail = lambda t: (a_d.data_ptr(), i_d.data_ptr(), c_d.data_ptr(), sc_d.data_ptr(), sz_d.data_ptr(), B, k) + tuple(x.data_ptr() for x in t[:6])
    assert lib.effdet_detections_hard(st, 0, l_d.data_ptr(), bsrc.data_ptr(), N if gather else 0, *tail(one), 0.3, 100, sc_d.data_ptr(),
                                      one[6].data_ptr(), one[7].data_ptr(), one[8].data_ptr()) == 0
    if gather:
        assert lib.effdet_decode_threshold_gather(st, 0, l_d.data_ptr(), bsrc.data_ptr(), N, *tail(two)) == 0
    else:
        assert lib.effdet_decode_threshold(st, 0, l_d.data_ptr(), bsrc.data_ptr(), *tail(two)) == 0
    assert lib.effdet_nms_hard(st, *[x.data_ptr() for x in two[:6]], B, k, 0.3, 100, sc_d.data_ptr(), two[6].data_ptr(), two[7].data_ptr(),
                               two[8].data_ptr()) == 0
    torch.cuda.synchronize()
    assert int(one[7].min()) > 0
    cnt = one[4].cpu()
    assert torch.equal(cnt, two[4].cpu())
    for b in range(B):                                   # the compacted candidates (only the first count[b] rows are defined)
        n = int(cnt[b])
        for x, y in zip(one[:4], two[:4]):
            assert torch.equal(x[b, :n].cpu(), y[b, :n].cpu())
    for x, y in zip(one[5:], two[5:]):
        assert torch.equal(x.cpu(), y.cpu())


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_soft_nms_golden(golden, tag):
    from ood_object_detection_amd.effdet.soft_nms import soft_nms, batched_soft_nms
    g = golden('soft_nms')
    boxes, scores, classes = (torch.from_numpy(g[tag + s]).to(DEV) for s in ('_boxes', '_scores', '_classes'))
    for fn, args, key in ((soft_nms, dict(method_gaussian=True), '_g'), (soft_nms, dict(method_gaussian=False), '_l')):
        i, s = fn(boxes, scores, sigma=0.5, iou_threshold=0.3, score_threshold=0.001, **args)
        assert i.numel() == len(g[tag + key + '_idx'])                 # every pick, like soft_nms.py:88-112
        assert np.array_equal(i.cpu().numpy(), g[tag + key + '_idx'])
        assert np.abs(s.cpu().numpy() - g[tag + key + '_scores']).max() <= 1e-6
    i, s = batched_soft_nms(boxes, scores, classes, method_gaussian=True, iou_threshold=0.3, score_threshold=0.001)
    assert i.numel() == len(g[tag + '_bg_idx'])
    assert np.array_equal(i.cpu().numpy(), g[tag + '_bg_idx'])
    assert np.abs(s.cpu().numpy() - g[tag + '_bg_scores']).max() <= 1e-6


@pytest.mark.parametrize('n', [1500, 9000])
def test_soft_nms_standalone_returns_every_pick_at_any_size(n):
    """the stand-alone API has no pick budget and no size limit (reference: soft_nms.py:88-112): n = 1500 (> the old 512-pick cap,
    register kernel) and n = 9000 (> 8192: the scratch-row kernel) against the oracle, run to exhaustion"""
    from ood_object_detection_amd.effdet.soft_nms import batched_soft_nms
    rs = np.random.RandomState(n)
    c = rs.uniform(0, 400, (n, 2)); wh = rs.uniform(8, 80, (n, 2))
    boxes = torch.from_numpy(np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32))
    scores = torch.from_numpy(rs.permutation(n).astype(np.float32) / n * 0.98 + 0.011)        # distinct
    classes = torch.from_numpy(rs.randint(0, 5, n).astype(np.int64))
    ri, rsc = op.batched_soft_nms(boxes, scores, classes, True, 0.5, 0.3, 0.001)
    i, s = batched_soft_nms(boxes.to(DEV), scores.to(DEV), classes.to(DEV), method_gaussian=True, sigma=0.5, iou_threshold=0.3, score_threshold=0.001)
    assert i.numel() == ri.numel() and ri.numel() > 512
    # rescored values differ by exp ulps; a swap of two picks needs two decayed scores within ~1e-7 of each other
    same = (i.cpu() == ri)
    assert float(same.float().mean()) >= 0.999
    assert float((s.cpu()[same] - rsc[same]).abs().max()) <= 1e-6


def test_nms_many_candidates_vs_oracle():
    """5000 candidates, 90 classes: hard + soft against the oracle, with and without img_info"""
    from ood_object_detection_amd.effdet.anchors import batched_detections
    anchors = op.anchor_boxes(3, 7, 3, [(1.0, 1.0), (1.4, 0.7), (0.7, 1.4)], 4.0, (256, 256))
    B, k, C = 2, 5000, 90
    rs = np.random.RandomState(3)
    idx = torch.from_numpy(rs.randint(0, anchors.shape[0], (B, k)).astype(np.int64))
    cls_id = torch.from_numpy(rs.randint(0, C, (B, k)).astype(np.int64))
    logits = torch.sort(torch.from_numpy(rs.normal(-3.0, 2.0, (B, k)).astype(np.float32)), dim=1, descending=True)[0]
    box = torch.from_numpy(rs.normal(0, 0.3, (B, k, 4)).astype(np.float32))
    scale, size = torch.tensor([1.3, 0.8]), torch.tensor([[300., 280.], [200., 210.]])
    for soft in (False, True):
        for info in (False, True):
            det, count, keep = batched_detections(logits.to(DEV), box.to(DEV), anchors.to(DEV), idx.to(DEV), cls_id.to(DEV),
                                                  scale.to(DEV) if info else None, size.to(DEV) if info else None, 100, soft)
            for b in range(B):
                ref, src = op.generate_detections(logits[b, :, None], box[b], anchors, idx[b], cls_id[b],
                                                  scale[b] if info else None, size[b] if info else torch.tensor(256), 100, soft,
                                                  return_aux=True)
                n = int(count[b])
                assert n == ref.shape[0]
                got = det[b, :n].cpu()
                assert torch.equal(got[:, 5], ref[:, 5])
                assert float((got[:, :4] - ref[:, :4]).abs().max()) <= 2e-4
                assert float((got[:, 4] - ref[:, 4]).abs().max()) <= 1e-6
                assert torch.equal(keep[b, :n].cpu().long(), src)
                assert float(det[b, n:].abs().sum()) == 0.0


def test_empty_and_edge_cases():
    from ood_object_detection_amd.effdet.anchors import batched_detections
    anchors = op.anchor_boxes(3, 7, 3, [(1.0, 1.0), (1.4, 0.7), (0.7, 1.4)], 4.0, (128, 128)).to(DEV)
    B, k = 2, 300
    logits = torch.full((B, k), -9.0, device=DEV)        # nothing passes 0.01
    logits[1, :3] = 3.0                                  # image 1: three identical boxes -> one survives hard NMS
    box = torch.zeros(B, k, 4, device=DEV)
    idx = torch.zeros(B, k, dtype=torch.int64, device=DEV)
    cls_id = torch.zeros(B, k, dtype=torch.int64, device=DEV)
    det, count, keep = batched_detections(logits, box, anchors, idx, cls_id, None, None, 100, False)
    assert count.tolist() == [0, 1] and float(det[0].abs().sum()) == 0.0
    det, count, keep = batched_detections(logits, box, anchors, idx, cls_id, None, None, 100, True)
    assert count.tolist() == [0, 3]                      # soft-NMS rescales duplicates instead of dropping them


# ------------------------------------------------------------------------------------ training-side ops
@pytest.mark.parametrize('tag,alpha,w,ls', [('pre', 0.15, 50.0, 0.0), ('inf', 0.25, 5.0, 0.0), ('ls', 0.25, 5.0, 0.1)])
def test_detection_loss_golden(golden, tag, alpha, w, ls):
    """loss values and autograd gradients vs the reference's loss_fn (effdet/loss.py:224-298)"""
    from _seeded import seeded_array
    from ood_object_detection_amd.effdet.loss import loss_fn
    g = golden('loss')
    B, C, A = [int(v) for v in g['meta'][:3]]
    sizes = [int(v) for v in g['meta'][3:]]
    cls_out = [torch.from_numpy(seeded_array(4, 'c%d' % i, (B, A * C, s, s), scale=1.5)).to(DEV).requires_grad_() for i, s in enumerate(sizes)]
    box_out = [torch.from_numpy(seeded_array(4, 'b%d' % i, (B, A * 4, s, s), scale=0.3)).to(DEV).requires_grad_() for i, s in enumerate(sizes)]
    cls_t = [torch.from_numpy(g['cls_t%d' % i]).to(DEV) for i in range(5)]
    box_t = [torch.from_numpy(g['box_t%d' % i]).to(DEV) for i in range(5)]
    npos = torch.from_numpy(g['npos']).to(DEV)
    total, cl, bl = loss_fn(cls_out, box_out, cls_t, box_t, npos, num_classes=C, alpha=alpha, gamma=1.5, delta=0.1,
                            box_loss_weight=w, label_smoothing=ls)
    ref = g[tag + '_loss']
    got = torch.stack([total, cl, bl]).detach().cpu().numpy()
    assert np.allclose(got, ref, rtol=2e-5, atol=1e-6), (got, ref)
    grads = torch.autograd.grad(total, cls_out + box_out)
    for i in range(5):
        assert np.allclose(grads[i].cpu().numpy(), g['%s_gc%d' % (tag, i)], rtol=1e-4, atol=1e-7)
        assert np.allclose(grads[5 + i].cpu().numpy(), g['%s_gb%d' % (tag, i)], rtol=1e-4, atol=1e-7)


def test_detection_loss_mixed_dtypes():
    """bfloat16 class outputs next to float32 box outputs (what a bf16 inference model returns): the loss upcasts the class
    outputs (the boxes keep their precision) and hands each input a gradient of its own dtype"""
    from _seeded import seeded_array
    from ood_object_detection_amd.effdet.loss import loss_fn
    B, C, A, sizes = 2, 6, 9, [8, 4, 2, 1, 1]
    cls32 = [torch.from_numpy(seeded_array(14, 'c%d' % i, (B, A * C, s, s), scale=1.5)).to(DEV).to(torch.bfloat16).float() for i, s in enumerate(sizes)]
    box32 = [torch.from_numpy(seeded_array(14, 'b%d' % i, (B, A * 4, s, s), scale=0.3)).to(DEV) for i, s in enumerate(sizes)]
    g = torch.Generator().manual_seed(3)
    cls_t = [torch.randint(-2, C, (B, s, s, A), generator=g).to(DEV) for s in sizes]
    box_t = [torch.randn(B, s, s, A * 4, generator=g).to(DEV) for s in sizes]
    npos = torch.tensor([3.0, 5.0], device=DEV)
    kw = dict(num_classes=C, alpha=0.25, gamma=1.5, delta=0.1, box_loss_weight=50.0)
    a = [t.clone().requires_grad_() for t in cls32]
    b = [t.clone().requires_grad_() for t in box32]
    ref, _, _ = loss_fn(a, b, cls_t, box_t, npos, **kw)
    gr = torch.autograd.grad(ref, a + b)
    a16 = [t.to(torch.bfloat16).requires_grad_() for t in cls32]
    b2 = [t.clone().requires_grad_() for t in box32]
    tot, _, _ = loss_fn(a16, b2, cls_t, box_t, npos, **kw)
    assert float(tot) == float(ref)
    gm = torch.autograd.grad(tot, a16 + b2)
    for i in range(5):
        assert gm[i].dtype == torch.bfloat16 and gm[5 + i].dtype == torch.float32
        assert torch.equal(gm[5 + i], gr[5 + i])
        assert torch.equal(gm[i], gr[i].to(torch.bfloat16))


def test_anchor_labeler_golden(golden):
    """class / box targets, matches and num_positives vs the reference's TargetAssigner.assign"""
    from ood_object_detection_amd import _lib
    from ood_object_detection_amd.effdet.anchors import Anchors, AnchorLabeler
    import _hip
    g = golden('labeler')
    anchors = Anchors(3, 7, 3, [(1.0, 1.0), (1.4, 0.7), (0.7, 1.4)], 4.0, (128, 128)).to(DEV)
    lab = AnchorLabeler(anchors, num_classes=6, match_threshold=0.5)
    gt_boxes = [torch.from_numpy(g['gt_boxes%d' % i]) for i in range(4)]
    gt_cls = [torch.from_numpy(g['gt_cls%d' % i]) for i in range(4)]
    cls_l, box_l, npos = lab.batch_label_anchors(gt_boxes, gt_cls)
    assert np.array_equal(npos.cpu().numpy(), g['npos'])
    N = anchors.boxes.shape[0]
    cls_flat = torch.cat([c.reshape(4, -1) for c in cls_l], 1).cpu().numpy()
    box_flat = torch.cat([b.reshape(4, -1, 4) for b in box_l], 1).cpu().numpy()
    for i in range(4):
        assert np.array_equal(cls_flat[i], g['cls_flat%d' % i])
        assert np.abs(box_flat[i] - g['box_flat%d' % i]).max() <= 2e-6
    # level shapes like the reference's unpack (anchors.py:421-433)
    assert [tuple(c.shape) for c in cls_l] == [(4, s, s, 9) for s in (16, 8, 4, 2, 1)]
    assert [tuple(b.shape) for b in box_l] == [(4, s, s, 36) for s in (16, 8, 4, 2, 1)]
    # raw C-ABI call: the match vector itself
    lib = _lib.load()
    B, Mmax = 4, 7
    gb = torch.zeros(B, Mmax, 4, device=DEV); gc = torch.full((B, Mmax), -1, dtype=torch.int64, device=DEV)
    for i in range(4):
        m = gt_boxes[i].shape[0]
        if m:
            gb[i, :m] = gt_boxes[i].to(DEV); gc[i, :m] = gt_cls[i].to(DEV)
    cls_t = torch.empty(B, N, dtype=torch.int64, device=DEV); box_t = torch.empty(B, N, 4, device=DEV)
    npos2 = torch.empty(B, device=DEV); match = torch.empty(B, N, dtype=torch.int64, device=DEV)
    nb = lib.effdet_label_anchors_workspace_bytes(B, Mmax, N)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    af = anchors.boxes.float().contiguous()
    rc = lib.effdet_label_anchors(_hip.stream(DEV), af.data_ptr(), gb.data_ptr(), gc.data_ptr(), B, Mmax, N, 0.5, cls_t.data_ptr(),
                                  box_t.data_ptr(), npos2.data_ptr(), match.data_ptr(), ws.data_ptr(), nb)
    assert rc == 0
    for i in range(4):
        assert np.array_equal(match[i].cpu().numpy(), g['match%d' % i])


def test_det_bench_train_runs():
    from _models import seeded_model
    from ood_object_detection_amd.effdet.bench import DetBenchTrain
    model, cfg, nodes, sd = seeded_model('tf_efficientdet_d0', 128, 6, seed=8)
    bench = DetBenchTrain(model.to(DEV).float(), create_labeler=True)
    bench.anchors.to(DEV)
    bench.anchor_labeler.anchors = bench.anchors.to(DEV)
    bench.eval()
    x = torch.randn(2, 3, 128, 128, device=DEV)
    target = {'bbox': [torch.tensor([[10., 12., 60., 70.], [30., 40., 100., 90.]]), torch.tensor([[5., 5., 50., 120.]])],
              'cls': [torch.tensor([1, 3]), torch.tensor([6])]}
    out = bench(x, target)
    assert all(bool(torch.isfinite(out[k]).all()) for k in ('loss', 'class_loss', 'box_loss'))
    assert out['detections'].shape == (2, 100, 6)
    assert abs(float(out['loss']) - float(out['class_loss']) - cfg.box_loss_weight * float(out['box_loss'])) < 1e-4 * max(1.0, float(out['loss']))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape', [(2, 3, 32, 48), (1, 3, 7, 9)])
def test_normalize_u8(dtype, shape):
    """effdet_normalize_u8 vs the oracle's restatement of PrefetchLoader (bit-exact in fp32; bf16 = rounded fp32)."""
    from oracle import preprocess as opre
    from ood_object_detection_amd.effdet.preprocess import normalize_batch
    g = torch.Generator().manual_seed(5)
    x = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)
    ref = torch.from_numpy(opre.normalize_u8(x.numpy()))
    got = normalize_batch(x.to(DEV), dtype=dtype)
    assert torch.equal(got.float().cpu(), ref.to(dtype).float())


def test_ood_image_score_and_auroc():
    """Device-side image score max_a(-energy) and pair-counting AUROC vs the oracle (exact)."""
    from ood_object_detection_amd import ood
    g = torch.Generator().manual_seed(11)
    e_in = torch.randn(37, 3069, generator=g) - 0.4
    e_out = torch.randn(29, 3069, generator=g)
    s_in, s_out = ood.image_scores(e_in.to(DEV)), ood.image_scores(e_out.to(DEV))
    assert np.array_equal(s_in.cpu().numpy(), op.image_ood_score(e_in.numpy()))
    assert np.array_equal(s_out.cpu().numpy(), op.image_ood_score(e_out.numpy()))
    assert abs(ood.auroc(s_in, s_out) - op.auroc(s_in.cpu().numpy(), s_out.cpu().numpy())) < 1e-12
    # heavy ties
    a = torch.round(torch.randn(500, generator=g) * 2) / 2
    b = torch.round(torch.randn(300, generator=g) * 2) / 2 - 0.5
    assert abs(ood.auroc(a.to(DEV), b.to(DEV)) - op.auroc(a.numpy(), b.numpy())) < 1e-12


def test_flat_adam_matches_oracle():
    """effdet_sqnorm + effdet_adam_clip_step (FlatAdam) vs the oracle's clip_grad_norm_ + Adam restatement."""
    from oracle import train as otr
    from ood_object_detection_amd.optim import FlatAdam
    g0 = torch.Generator().manual_seed(8)
    shapes = [(64, 40), (1000,), (8, 3, 3, 3), (100003,)]
    params = [torch.nn.Parameter(torch.randn(*s, generator=g0).to(DEV)) for s in shapes]
    p = np.concatenate([q.detach().cpu().numpy().ravel() for q in params]); m = np.zeros_like(p); v = np.zeros_like(p)
    opt = FlatAdam(params, lr=1e-3, max_grad_norm=10.0)
    for step in range(1, 4):
        grads = [torch.randn(*s, generator=g0) * (0.5 if step == 2 else 0.001) for s in shapes]    # step 2 clips
        opt.zero_grad()
        for q, g in zip(params, grads):
            q.grad.add_(g.to(DEV))                    # accumulate the way autograd does: in place, into the flat buffer
        norm = opt.step()
        p, m, v, n = otr.clip_adam_step(p, np.concatenate([g.numpy().ravel() for g in grads]), m, v, step)
        got = np.concatenate([q.detach().cpu().numpy().ravel() for q in params])
        assert abs(float(norm) - float(n)) < 1e-5 * max(1.0, float(n))
        assert np.abs(got - p).max() < 2e-6
        # parameters sit on 64-byte boundaries of the flat buffers; the padding stays zero
        off, ma, va = 0, [], []
        for q in params:
            ma.append(opt.exp_avg[off:off + q.numel()])
            va.append(opt.exp_avg_sq[off:off + q.numel()])
            assert q.data_ptr() % 64 == 0
            off += opt._padded(q.numel())
        assert np.abs(torch.cat(ma).cpu().numpy() - m).max() < 1e-6 and np.abs(torch.cat(va).cpu().numpy() - v).max() < 1e-6


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_topk_full_size_properties(dtype):
    """BASELINE size (640 px, C = 90: 6.9 M logits per image), checked through size-independent properties:
    values sorted descending, ties in ascending flat index, each value is the logit at its index, and the value
    multiset equals torch.topk's (the oracle's Python loop would take minutes here)."""
    from ood_object_detection_amd.effdet.bench import _post_process
    B, C, A, k = 2, 90, 9, 5000
    sizes = [80, 40, 20, 10, 5]
    g = torch.Generator(device=DEV).manual_seed(21)
    cls = [(torch.randn(B, A * C, s, s, device=DEV, generator=g) * 0.7 - 3.0).to(dtype) for s in sizes]
    box = [torch.randn(B, A * 4, s, s, device=DEV, generator=g).to(dtype) for s in sizes]
    flat = torch.cat([c.permute(0, 2, 3, 1).reshape(B, -1, C) for c in cls], 1)
    amax = flat.float().max(dim=2).values
    for am in (None, amax):
        vals, boxes, idx, cid = _post_process(cls, box, 5, C, k, anchor_max=am)
        v = vals.reshape(B, k).float()
        fidx = idx * C + cid
        assert torch.all(v[:, :-1] >= v[:, 1:])
        tie = v[:, :-1] == v[:, 1:]
        assert torch.all(fidx[:, :-1][tie] < fidx[:, 1:][tie])
        assert torch.equal(v, flat.reshape(B, -1).float().gather(1, fidx))
        ref = torch.topk(flat.reshape(B, -1).float(), k, dim=1).values
        assert torch.equal(v, ref)
        bflat = torch.cat([b.permute(0, 2, 3, 1).reshape(B, -1, 4) for b in box], 1)
        assert torch.equal(boxes, bflat.gather(1, idx.unsqueeze(-1).expand(-1, -1, 4)))


@pytest.mark.parametrize('hw', [(480, 640), (333, 500), (64, 48), (100, 37), (128, 128), (720, 1280)])
def test_resize_pad_u8(hw):
    """Device ResizePad vs the oracle (itself pinned bit-exactly against PIL): bit exact, CHW output."""
    from oracle import preprocess as opre
    from ood_object_detection_amd.effdet.preprocess import resize_pad, resolve_fill_color
    rng = np.random.RandomState(hw[0] * 7 + hw[1])
    img = rng.randint(0, 256, (hw[0], hw[1], 3)).astype(np.uint8)
    fill = resolve_fill_color('mean')
    ref, rs = opre.resize_pad(img, 128, fill)
    got, gs = resize_pad(torch.from_numpy(img).to(DEV), 128, fill)
    assert gs == rs
    assert np.array_equal(got.cpu().numpy(), np.transpose(ref, (2, 0, 1)))


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_detection_evaluator_golden(golden, tag):
    """device evaluator (effdet_eval_match + effdet_eval_ap) vs the reference's ObjectDetectionEvaluator fixture, through the
    reference's per-image API and through the batched native entry"""
    from _seeded import eval_case
    from ood_object_detection_amd.effdet.evaluation import ObjectDetectionEvaluator
    g = golden('evaluation')
    seed, n_img, C, n_det = [int(v) for v in g[tag + '_meta']]
    images = eval_case(seed, n_img, C, n_det)
    cats = [{'id': i + 1, 'name': 'c%d' % i} for i in range(C)]
    names = ['c%d' % i for i in range(C)]

    def check(m):
        assert abs(m['Precision/mAP@0.5IOU'] - float(g[tag + '_map'])) < 1e-12
        assert abs(m['Precision/meanCorLoc@0.5IOU'] - float(g[tag + '_corloc'])) < 1e-12
        assert np.allclose([m['AP@0.5IOU/c%d' % i] for i in range(C)], g[tag + '_ap'], rtol=0, atol=1e-12, equal_nan=True)
        assert np.allclose([m['CorLoc@0.5IOU/c%d' % i] for i in range(C)], g[tag + '_cl'], rtol=0, atol=1e-12, equal_nan=True)

    ev = ObjectDetectionEvaluator(cats, evaluate_corlocs=True, device=DEV)
    for i, im in enumerate(images):
        ev.add_single_ground_truth_image_info(i, {'bbox': im['gt_boxes'], 'cls': im['gt_classes']})
        ev.add_single_detected_image_info(i, {'bbox': im['det_boxes'], 'scores': im['det_scores'], 'cls': im['det_classes']})
    check(ev.evaluate(names))
    # batched: one launch for all images, rows in the DetBenchPredict layout (xyxy, score, class), padded ground truth
    ev.clear()
    B, M = len(images), max(1, max(len(im['gt_classes']) for im in images))
    det = torch.zeros(B, n_det + 3, 6)
    cnt = torch.zeros(B, dtype=torch.int32)
    gtb, gtc = torch.zeros(B, M, 4), torch.full((B, M), -1, dtype=torch.int64)
    for i, im in enumerate(images):
        b = torch.from_numpy(im['det_boxes'])
        det[i, :n_det] = torch.stack([b[:, 1], b[:, 0], b[:, 3], b[:, 2], torch.from_numpy(im['det_scores']),
                                      torch.from_numpy(im['det_classes']).float()], 1)
        cnt[i] = n_det
        m = len(im['gt_classes'])
        gtb[i, :m] = torch.from_numpy(im['gt_boxes'])
        gtc[i, :m] = torch.from_numpy(im['gt_classes'])
    ev.add_batch(det.to(DEV), cnt.to(DEV), gtb.to(DEV), gtc.to(DEV))
    check(ev.evaluate(names))
    ev.clear()
    m = ev.evaluate(names)
    assert np.isnan(m['Precision/mAP@0.5IOU'])
    # ground truth of an image that never gets detections still counts in the recall denominator
    from oracle import evaluation as oe
    ev.clear()
    ims = []
    for i, im in enumerate(images):
        ev.add_single_ground_truth_image_info(i, {'bbox': im['gt_boxes'], 'cls': im['gt_classes']})
        if i % 2 == 0:
            ev.add_single_detected_image_info(i, {'bbox': im['det_boxes'], 'scores': im['det_scores'], 'cls': im['det_classes']})
            ims.append(dict(det_boxes=im['det_boxes'], det_scores=im['det_scores'], det_classes=im['det_classes'] - 1,
                            gt_boxes=im['gt_boxes'], gt_classes=im['gt_classes'] - 1))
        else:
            ims.append(dict(det_boxes=np.zeros((0, 4), np.float32), det_scores=np.zeros(0, np.float32), det_classes=np.zeros(0, np.int64),
                            gt_boxes=im['gt_boxes'], gt_classes=im['gt_classes'] - 1))
    with np.errstate(all='ignore'):
        r = oe.evaluate(ims, C)
    m = ev.evaluate(names)
    assert (np.isnan(r['mean_ap']) and np.isnan(m['Precision/mAP@0.5IOU'])) or abs(m['Precision/mAP@0.5IOU'] - r['mean_ap']) < 1e-12
    assert np.allclose([m['AP@0.5IOU/c%d' % i] for i in range(C)], r['per_class_ap'], rtol=0, atol=1e-12, equal_nan=True)


@pytest.mark.parametrize('target', ['avg', 'max'])
def test_novelty_score_matches_infer_py_expressions(target):
    """the fork's soft_thresh * similarity score (infer.py:425-427, 465-471, 607-616) against the literal torch expressions"""
    from ood_object_detection_amd import ood
    g = torch.Generator().manual_seed(8)
    n, d, m = 3000, 64, 25
    e = torch.randn(n, d, generator=g) * 3.0
    e[5] = 0.0                                        # a zero row: F.normalize's eps path
    conf = torch.randn(n, generator=g) * 2.0 - 3.0
    proto = torch.randperm(n, generator=g)[:m]
    ref, st, sim = op.novelty_score(e, conf, proto, 3.0, 3.0, target)
    out = ood.novelty_score(e.to(DEV), conf.to(DEV), proto.to(DEV), 3.0, 3.0, target)
    assert float((out['soft_thresh'].cpu() - st).abs().max()) <= 1e-6
    assert float((out['sim'].cpu() - sim).abs().max()) <= 2e-6
    assert float((out['score'].cpu() - ref).abs().max()) <= 2e-6



def test_novelty_score_rejects_out_of_range_prototypes():
    """a prototype index outside [0, n) (a stale max_idxs, an index into another concatenation) is not dereferenced: scores come
    back NaN instead of a device fault or silently wrong prototypes"""
    from ood_object_detection_amd import ood
    n, d = 300, 128
    e = _rand(n, d, seed=5).to(DEV)
    conf = _rand(n, seed=6).to(DEV)
    good = ood.novelty_score(e, conf, torch.tensor([0, 7, n - 1]), 3.0, 3.0, 'avg')
    assert bool(torch.isfinite(good['score']).all())
    for bad in ([0, n, 7], [-1, 3], [2 ** 40]):
        out = ood.novelty_score(e, conf, torch.tensor(bad), 3.0, 3.0, 'max')
        torch.cuda.synchronize()
        assert bool(torch.isnan(out['score']).all()) and bool(torch.isnan(out['sim']).all())
        assert bool(torch.isfinite(out['soft_thresh']).all())
