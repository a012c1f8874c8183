"""Minimal forward loop for profiling: python3 tools/run_forward.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as B
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
model = B.build_model('tf_efficientdet_d0', 640, 90).to('cuda:0').to(torch.bfloat16)
from ood_object_detection_amd.effdet.bench import DetBenchPredict
bench = DetBenchPredict(model).to('cuda:0')
x = torch.randn(64, 3, 640, 640, device='cuda:0').to(torch.bfloat16)
with torch.no_grad():
    for _ in range(steps):
        bench(x)
torch.cuda.synchronize()
print('done')
