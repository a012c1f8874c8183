"""Probe: does running two half-batches on two streams (inside one hipGraph) beat one full batch?"""
import os, sys, copy, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as B
from ood_object_detection_amd.effdet.bench import DetBenchPredict
dev = torch.device('cuda:0')
model = B.build_model('tf_efficientdet_d0', 640, 90).to(dev).to(torch.bfloat16)
x = torch.randn(64, 3, 640, 640, device=dev).to(torch.bfloat16)
sizes = [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else [32, 32]
nsplit = len(sizes)
benches = [DetBenchPredict(copy.deepcopy(model) if i else model, streams=1).to(dev) for i in range(nsplit)]
xs = list(x.split(sizes))
streams = [torch.cuda.Stream(dev) for _ in range(nsplit)]
def step():
    cur = torch.cuda.current_stream(dev)
    outs = []
    for b, xi, s in zip(benches, xs, streams):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(b(xi))
    for s in streams:
        cur.wait_stream(s)
    return outs
with torch.no_grad():
    for _ in range(3): step()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side): step()
    torch.cuda.current_stream(dev).wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
print('splits', sizes, 'ms/step', dt * 1e3, 'img/s', 64 / dt)
