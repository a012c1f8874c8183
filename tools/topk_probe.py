"""effdet_topk_select (anchor select + collect + sort: three launches) alone on the bench workload's logits: ms per call.
usage: topk_probe.py [reps]   EFFDET_LIB_VARIANT=libeffdet_hip_<tag>.so selects an A/B build (tools only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ood_object_detection_amd import _lib
if os.environ.get('EFFDET_LIB_VARIANT'):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ['EFFDET_LIB_VARIANT'])
import bench as Bn
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lib = _lib.load()
dev = 'cuda:0'
model = Bn.build_model('tf_efficientdet_d0', 640, 90).to(dev).to(torch.bfloat16)
B = 64
x = torch.randn(B, 3, 640, 640, device=dev).to(torch.bfloat16)
with torch.no_grad():
    model(x)
eng = model._engine
N, C, k = eng.N, eng.C, 5000
cls_all = eng.cls_all
ws_bytes = lib.effdet_topk_workspace_bytes(B, N)
ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=dev)
oc = torch.empty(B, k, 1, dtype=cls_all.dtype, device=dev)
idx = torch.empty(B, k, dtype=torch.int64, device=dev); cid = torch.empty_like(idx)
st = torch.cuda.current_stream().cuda_stream


def run():
    rc = lib.effdet_topk_select(st, 1, cls_all.data_ptr(), eng.ood_max_logit.data_ptr(), B, N, C, None, k,
                                oc.data_ptr(), None, idx.data_ptr(), cid.data_ptr(), ws.data_ptr(), ws_bytes)
    assert rc == 0, rc


run(); torch.cuda.synchronize()
ref_idx, ref_cid = idx.clone(), cid.clone()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record(); torch.cuda.synchronize()
print('topk_select B=%d N=%d C=%d k=%d: %.4f ms per call  (checksum %d %d)' % (B, N, C, k, e0.elapsed_time(e1) / reps, int(ref_idx.sum()), int(ref_cid.sum())))
