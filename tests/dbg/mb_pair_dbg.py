import sys, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import _hip
from ood_object_detection_amd import _lib, pairfmt
lib = _lib.load()
DEV = 'cuda:0'
torch.manual_seed(0)
B, Cin, mid, H, W, k, s = 1, 80, 480, 33, 40, 3, 2
x = torch.randn(B, H, W, Cin)
w1 = torch.randn(mid, Cin) * 0.1
taps = torch.randn(k * k, mid) * 0.3
one, zero = torch.ones(mid), torch.zeros(mid)
Ho, Wo = (H + 1) // 2, (W + 1) // 2
y = pairfmt.encode(torch.full((B, Ho, Wo, mid), 777.0)).to(DEV)
nt = lib.effdet_mbconv_tiles_per_image(2, H, W, Cin, mid, k, s)
part = torch.zeros(B, nt, mid, device=DEV)
keep = [pairfmt.encode(x).to(DEV), pairfmt.encode(w1).to(DEV), one.to(DEV), zero.to(DEV), taps.to(DEV), one.clone().to(DEV), zero.clone().to(DEV)]
rc = lib.effdet_mbconv_expand_dw(_hip.stream(DEV), 2, keep[0].data_ptr(), y.data_ptr(), *[t.data_ptr() for t in keep[1:]], part.data_ptr(), B, H, W, Cin, mid, k, s)
torch.cuda.synchronize()
print('rc', rc, 'parts', nt)
got = pairfmt.decode(y.cpu())
bad = torch.isnan(got) | (got == 777.0)
print('nan', int(torch.isnan(got).sum()), 'unwritten', int((got == 777.0).sum()))
print('nan count', int(bad.sum()), 'of', bad.numel())
idx = bad.nonzero()
print('rows with nan', sorted(set(idx[:, 1].tolist())))
print('cols with nan', sorted(set(idx[:, 2].tolist())))
print('channels with nan (first 40)', sorted(set(idx[:, 3].tolist()))[:40])
print('raw y nan?', int(torch.isnan(y).sum()))

xq = pairfmt.decode(pairfmt.encode(x)).permute(0, 3, 1, 2).double()
import torch.nn.functional as F
e = F.conv2d(xq, pairfmt.decode(pairfmt.encode(w1)).double()[:, :, None, None]); e = e * torch.sigmoid(e)
ep = F.pad(e, (0, 1, 1, 1))   # W: pad_l 0, pad_r 1 ; H: 33 -> pad_t 1, pad_b 1
ref = F.conv2d(ep, taps.t().reshape(mid, 1, 3, 3).double(), stride=2, groups=mid); ref = ref * torch.sigmoid(ref)
d = (got.permute(0, 3, 1, 2).double() - ref).abs()
d[torch.isnan(d)] = 1e9
print('max err excluding', float(d[d < 1e8].max()), 'shape', tuple(ref.shape))
for (b_, r_, c_, ch_) in idx[:3].tolist():
    print('at', r_, c_, ch_, 'got', float(got[b_, r_, c_, ch_]), 'ref', float(ref[b_, ch_, r_, c_]))
