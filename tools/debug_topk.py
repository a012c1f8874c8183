"""Debug helper: candidate counts of the two-stage top-k on the bench workload."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, numpy as np
import bench as Bn
from ood_object_detection_amd import _lib
lib = _lib.load()
model = Bn.build_model('tf_efficientdet_d0', 640, 90).to('cuda:0').to(torch.bfloat16)
x = torch.randn(8, 3, 640, 640, device='cuda:0').to(torch.bfloat16)
with torch.no_grad():
    model(x)
eng = model._engine
B, N, C, k = 8, eng.N, eng.C, 5000
cls_all, box_all = eng.cls_all[:B], eng.box_all[:B]
ws_bytes = lib.effdet_topk_workspace_bytes(B, N)
ws = torch.zeros(ws_bytes, dtype=torch.uint8, device='cuda:0')
oc = torch.empty(B, k, 1, dtype=cls_all.dtype, device='cuda:0'); ob = torch.empty(B, k, 4, dtype=cls_all.dtype, device='cuda:0')
idx = torch.empty(B, k, dtype=torch.int64, device='cuda:0'); cid = torch.empty_like(idx)
st = torch.cuda.current_stream().cuda_stream
rc = lib.effdet_topk_select(st, 1, cls_all.data_ptr(), eng.ood_max_logit.data_ptr(), B, N, C, box_all.data_ptr(), k,
                            oc.data_ptr(), ob.data_ptr(), idx.data_ptr(), cid.data_ptr(), ws.data_ptr(), ws_bytes)
torch.cuda.synchronize()
s = ws[:2 * B * 64].cpu().numpy().view(np.uint32).reshape(2, B, 16)
for name, a in (('stage1', s[0]), ('stage2', s[1])):
    print(name, 'bits_done', a[:, 2], 'done', a[:, 3], 'c_hi', a[:, 4], 'cand_total', a[:, 5], 'cand_count', a[:, 6])
am = eng.ood_max_logit[:B].float()
print('anchor max: min/median/max', am.min().item(), am.median().item(), am.max().item())
print('logits std', cls_all.float().std().item())
