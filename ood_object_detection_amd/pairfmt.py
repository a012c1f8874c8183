"""Host-side helpers for dtype 2 of the C ABI (EFFDET_BF16X2, the "accurate" mode): values stored as the unevaluated sum of two
bfloat16 numbers, x ~ hi + lo with hi = bf16(x), lo = bf16(x - hi) (16 significand bits).  Layout: the last dimension (channels,
or the K dimension of a conv weight) is cut into groups of 8; a group is 32 bytes, [8 x bf16 hi][8 x bf16 lo] - so the tensor
has the shape and byte size of its float32 counterpart and is carried in torch.float32 storage that must not be read as
floats.  Used for parameter-sized glue (packing weights when an engine is built) and by the tests; activation tensors are
encoded by the kernels that produce them."""
import torch


def encode(x):
    """float tensor [..., C] (C % 8 == 0) -> opaque float32-typed tensor of the same shape in the two-term layout"""
    if x.shape[-1] % 8:
        raise ValueError('the two-term bf16 layout needs a multiple of 8 values in the last dimension (got %d)' % x.shape[-1])
    x = x.detach().to(torch.float32).contiguous()
    hi = x.to(torch.bfloat16)
    lo = (x - hi.to(torch.float32)).to(torch.bfloat16)
    g = x.shape[:-1] + (x.shape[-1] // 8, 8)
    packed = torch.stack((hi.reshape(g), lo.reshape(g)), dim=-2).contiguous()       # [..., C/8, 2, 8] bf16
    return packed.view(torch.float32).reshape(x.shape)


def decode(t):
    """inverse of encode: opaque tensor [..., C] -> float32 values hi + lo (exact in float32)"""
    if t.dtype != torch.float32 or t.shape[-1] % 8:
        raise ValueError('expected float32-typed two-term storage with a multiple of 8 channels')
    t = t.contiguous()
    p = t.reshape(t.shape[:-1] + (t.shape[-1] // 8, 8)).view(torch.bfloat16)          # [..., C/8, 16] bf16
    p = p.reshape(t.shape[:-1] + (t.shape[-1] // 8, 2, 8)).to(torch.float32)
    return (p[..., 0, :] + p[..., 1, :]).reshape(t.shape)
