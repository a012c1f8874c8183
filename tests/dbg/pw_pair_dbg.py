import sys, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import _hip
from ood_object_detection_amd import _lib, pairfmt
lib = _lib.load()
DEV = 'cuda:0'
torch.manual_seed(0)
M, K, N = 64, 16, 32
A = pairfmt.decode(pairfmt.encode(torch.randn(M, K)))
W = pairfmt.decode(pairfmt.encode(torch.randn(N, K) * 0.25))
sh = torch.zeros(N)
out = torch.full((M, N), 7.0, device=DEV)
rc = lib.effdet_pw_gemm_bn_act(_hip.stream(DEV), 2, pairfmt.encode(A).to(DEV).data_ptr(), M, K, pairfmt.encode(W).to(DEV).data_ptr(), N, None,
                               sh.to(DEV).data_ptr(), 0, None, None, 0, out.data_ptr(), 0, 0)
torch.cuda.synchronize()
print('rc', rc)
ref = A.double() @ W.double().t()
got = pairfmt.decode(out.cpu())
print('raw out[0,:8] as float', out[0, :8].cpu())
print('got[0,:8]', got[0, :8])
print('ref[0,:8]', ref[0, :8])
print('got[5,8:16]', got[5, 8:16]); print('ref[5,8:16]', ref[5, 8:16])
print('err', float((got - ref).abs().max()), 'max ref', float(ref.abs().max()))
# bf16 path sanity
out2 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
rc = lib.effdet_pw_gemm_bn_act(_hip.stream(DEV), 1, A.to(torch.bfloat16).to(DEV).data_ptr(), M, K, W.to(torch.bfloat16).to(DEV).data_ptr(), N, None,
                               sh.to(DEV).data_ptr(), 0, None, None, 0, out2.data_ptr(), 0, 0)
torch.cuda.synchronize()
print('bf16 err', float((out2.float().cpu() - ref).abs().max()))
