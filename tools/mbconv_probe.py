"""One MBConv front launch (expand 1x1 + depthwise, the C ABI entry point) in isolation, for rocprofv3 / PMC passes.
usage: python3 tools/mbconv_probe.py H W Cin mid k s [B] [reps] [gated]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ood_object_detection_amd import _lib

H, W, Cin, mid, k, s = [int(v) for v in sys.argv[1:7]]
B = int(sys.argv[7]) if len(sys.argv) > 7 else 64
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 5
gated = len(sys.argv) > 9 and sys.argv[9] == 'gated'
lib = _lib.load()
dev = 'cuda:0'
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(B, H, W, Cin, device=dev, generator=g).to(torch.bfloat16)
w1 = (torch.randn(mid, Cin, device=dev, generator=g) * Cin ** -0.5).to(torch.bfloat16)
s1 = torch.rand(mid, device=dev, generator=g) + 0.5; t1 = torch.randn(mid, device=dev, generator=g) * 0.2
taps = torch.randn(k * k, mid, device=dev, generator=g) / k
s2 = torch.rand(mid, device=dev, generator=g) + 0.5; t2 = torch.randn(mid, device=dev, generator=g) * 0.2
Ho, Wo = (H + s - 1) // s, (W + s - 1) // s
y = torch.empty(B, Ho, Wo, mid, dtype=torch.bfloat16, device=dev)
nt = (lib.effdet_mbconv_gated_tiles_per_image if gated else lib.effdet_mbconv_tiles_per_image)(1, H, W, Cin, mid, k, s)
part = torch.zeros(B, nt, mid, dtype=torch.float32, device=dev)
gate = torch.rand(B, Cin, device=dev, generator=g)
st = torch.cuda.current_stream().cuda_stream


def run():
    if gated:
        rc = lib.effdet_mbconv_expand_dw_gated(st, 1, x.data_ptr(), gate.data_ptr(), y.data_ptr(), w1.data_ptr(), s1.data_ptr(), t1.data_ptr(),
                                               taps.data_ptr(), s2.data_ptr(), t2.data_ptr(), part.data_ptr(), B, H, W, Cin, mid, k, s)
    else:
        rc = lib.effdet_mbconv_expand_dw(st, 1, x.data_ptr(), y.data_ptr(), w1.data_ptr(), s1.data_ptr(), t1.data_ptr(),
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
nbytes = (x.numel() + y.numel()) * 2
print('mbconv H=%d W=%d Cin=%d mid=%d k=%d s=%d B=%d%s: %.4f ms  %.0f GB/s (in+out)  parts=%d' % (H, W, Cin, mid, k, s, B, ' gated' if gated else '', ms, nbytes / ms / 1e6, nt))
