"""Every MBConv front launch of a backbone's late stages in isolation: time (HIP events) and a parity check against the same
arithmetic in torch on the GPU (float32 conv ops; a checker for this tool only - the product never runs them).
usage: python3 tools/mbconv_layers.py [B] [reps] [shape ...]   shape = H,W,Cin,mid,k,s   (default: d0 / 640 blocks 3.0 ... 6.0)
MB_PAIR=1: the two-term (accurate) mode of the same launches (dtype 2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from ood_object_detection_amd import _lib, pairfmt

D0 = [('3.0', 80, 80, 40, 240, 3, 2), ('3.1', 40, 40, 80, 480, 3, 1), ('3.2', 40, 40, 80, 480, 3, 1), ('4.0', 40, 40, 80, 480, 5, 1),
      ('4.1', 40, 40, 112, 672, 5, 1), ('4.2', 40, 40, 112, 672, 5, 1), ('5.0', 40, 40, 112, 672, 5, 2), ('5.1', 20, 20, 192, 1152, 5, 1),
      ('5.2', 20, 20, 192, 1152, 5, 1), ('5.3', 20, 20, 192, 1152, 5, 1), ('6.0', 20, 20, 192, 1152, 3, 1)]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
shapes = [('arg%d' % i,) + tuple(int(v) for v in a.split(',')) for i, a in enumerate(sys.argv[3:])] or D0
if os.environ.get('EFFDET_LIB_VARIANT'):          # A/B and ablation builds (make variant TAG=..): tools only
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ['EFFDET_LIB_VARIANT'])
lib = _lib.load()
dev = 'cuda:0'
PAIR = os.environ.get('MB_PAIR') == '1'
DT = 2 if PAIR else 1
st = torch.cuda.current_stream().cuda_stream


def same_pad(x, k, s):
    H, W = x.shape[2], x.shape[3]
    ph = max((-(-H // s) - 1) * s + k - H, 0)
    pw = max((-(-W // s) - 1) * s + k - W, 0)
    return F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))


total = 0.0
for name, H, W, Cin, mid, k, s in shapes:
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(B, H, W, Cin, device=dev, generator=g).to(torch.bfloat16)
    w1 = (torch.randn(mid, Cin, device=dev, generator=g) * Cin ** -0.5).to(torch.bfloat16)
    xk, wk = (pairfmt.encode(x.float()), pairfmt.encode(w1.float())) if PAIR else (x, w1)      # what the kernel reads
    s1 = torch.rand(mid, device=dev, generator=g) + 0.5; t1 = torch.randn(mid, device=dev, generator=g) * 0.2
    taps = torch.randn(k * k, mid, device=dev, generator=g) / k
    s2 = torch.rand(mid, device=dev, generator=g) + 0.5; t2 = torch.randn(mid, device=dev, generator=g) * 0.2
    Ho, Wo = (H + s - 1) // s, (W + s - 1) // s
    y = torch.full((B, Ho, Wo, mid), float('nan'), dtype=torch.float32 if PAIR else torch.bfloat16, device=dev)
    nt = lib.effdet_mbconv_tiles_per_image(DT, H, W, Cin, mid, k, s)
    part = torch.full((B, nt, mid), float('nan'), dtype=torch.float32, device=dev)

    def run():
        rc = lib.effdet_mbconv_expand_dw(st, DT, xk.data_ptr(), y.data_ptr(), wk.data_ptr(), s1.data_ptr(), t1.data_ptr(),
                                         taps.data_ptr(), s2.data_ptr(), t2.data_ptr(), part.data_ptr(), B, H, W, Cin, mid, k, s)
        assert rc == 0, rc
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    total += ms
    # reference on a few images
    nb = min(B, 4)
    xf = x[:nb].float().permute(0, 3, 1, 2)
    e = F.conv2d(xf, w1.float()[:, :, None, None]) * s1[None, :, None, None] + t1[None, :, None, None]
    e = (e * torch.sigmoid(e))
    if not PAIR:
        e = e.to(torch.bfloat16).float()
    wd = taps.t().reshape(mid, 1, k, k)
    o = F.conv2d(same_pad(e, k, s), wd, stride=s, groups=mid) * s2[None, :, None, None] + t2[None, :, None, None]
    ref = o * torch.sigmoid(o)
    got = (pairfmt.decode(y[:nb]) if PAIR else y[:nb].float()).permute(0, 3, 1, 2)
    err = float((got - ref).abs().max() / ref.abs().max())
    pooled = part[:nb].sum(1) / (Ho * Wo)
    perr = float((pooled - ref.mean((2, 3))).abs().max())
    nbytes = (x.numel() + y.numel()) * (4 if PAIR else 2)
    print('%-5s H=%d W=%d Cin=%d mid=%d k=%d s=%d B=%d: %.4f ms  %6.0f GB/s  parts=%d  rel err %.4f  pool err %.5f %s' % (
        name, H, W, Cin, mid, k, s, B, ms, nbytes / ms / 1e6, nt, err, perr, 'OK' if err < (1e-4 if PAIR else 0.03) and perr < (1e-4 if PAIR else 5e-3) else 'MISMATCH'), flush=True)
print('sum %.4f ms' % total)
