"""Every project conv (pw_gemm launch) of the d0 backbone plan in isolation: HIP-event time per launch.
EFFDET_LIB_VARIANT=libeffdet_hip_<tag>.so selects an A/B build (tools only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ood_object_detection_amd import _lib
if os.environ.get('EFFDET_LIB_VARIANT'):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ['EFFDET_LIB_VARIANT'])
import bench
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda', 0)
model = bench.build_model('tf_efficientdet_d0', 640, 90).to(dev).to(torch.bfloat16)
x = torch.randn(64, 3, 640, 640, device=dev).to(torch.bfloat16)
with torch.no_grad():
    model(x)
eng = model._engine
st = torch.cuda.current_stream(dev)
tot = 0.0
for fn, args, what, meta in eng._bb_plan + eng._fpn_plan[:1]:
    if meta['kind'] != 'pw_gemm':
        continue
    fn(st.cuda_stream, *args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        fn(st.cuda_stream, *args)
    e1.record(st)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tot += ms
    print('%-34s %.4f ms  %.0f GB/s' % (what[:34], ms, meta['bytes'] / ms / 1e6))
print('sum %.4f ms' % tot)
