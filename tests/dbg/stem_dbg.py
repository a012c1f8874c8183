import sys, torch, ctypes
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import _hip
from oracle import model as om
from ood_object_detection_amd import _lib, pairfmt
lib = _lib.load()
DEV = 'cuda:0'
torch.manual_seed(0)
B, H, W, C = 1, 96, 128, 32
ws = torch.randn(C, 3, 3, 3) * 0.3
wd = torch.randn(C, 1, 3, 3) * 0.3
one, zero = torch.ones(C), torch.zeros(C)
wk = torch.zeros(C, 32); wk[:, :27] = ws.permute(0, 2, 3, 1).reshape(C, 27)
x = torch.randn(B, 3, H, W)
e = om.conv2d_pad(x.double(), ws.double(), None, 2, 'same'); e = e * torch.sigmoid(e)
ref = om.conv2d_pad(e, wd.double(), None, 1, 'same', groups=C); ref = ref * torch.sigmoid(ref)
for dt in (2, 1):
    Ho, Wo = ref.shape[2], ref.shape[3]
    nt = lib.effdet_stem_dw_parts(dt, H, W, C)
    y = torch.zeros(B, Ho, Wo, C, dtype=torch.float32 if dt == 2 else torch.bfloat16, device=DEV)
    part = torch.zeros(B, nt, C, device=DEV)
    keep = [(wk if dt == 2 else wk.to(torch.bfloat16)).to(DEV), one.to(DEV), zero.to(DEV), wd.permute(2, 3, 0, 1).reshape(9, C).contiguous().to(DEV), one.clone().to(DEV), zero.clone().to(DEV)]
    xd = x.to(DEV)
    rc = lib.effdet_stem_dw_fused(_hip.stream(DEV), 0, dt, xd.data_ptr(), *[t.data_ptr() for t in keep], y.data_ptr(), part.data_ptr(), B, H, W, C)
    torch.cuda.synchronize()
    got = (pairfmt.decode(y.cpu()) if dt == 2 else y.float().cpu()).permute(0, 3, 1, 2).double()
    d = (got - ref).abs()
    print('dtype', dt, 'rc', rc, 'max err', float(d.max()), 'max ref', float(ref.abs().max()))
    bad = (d > 0.05).nonzero()
    print(' bad count', len(bad), 'rows', sorted(set(bad[:, 2].tolist()))[:20], 'cols', sorted(set(bad[:, 3].tolist()))[:40])
