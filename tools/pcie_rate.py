#!/usr/bin/env python
"""PCIe-inclusive throughput of the bench workload: raw uint8 images start in pinned HOST memory (what a data loader hands
over), are copied to the GPU and run through DetBenchPredict (normalisation fused into the stem).  Reported next to the
resident-input number of bench.py; never bench.py's `value`.

  serial      copy -> forward -> copy -> forward ...          (one stream)
  overlapped  the copy of batch i+1 runs on a second stream while batch i computes (two device buffers)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def main():
    from bench import build_model
    from ood_object_detection_amd.effdet.bench import DetBenchPredict
    dev = torch.device('cuda', 0)
    B, S, steps = 64, 640, 20
    model = build_model('tf_efficientdet_d0', S, 90).to(dev).to(torch.bfloat16)
    bench = DetBenchPredict(model).to(dev)
    host = [torch.randint(0, 256, (B, 3, S, S), dtype=torch.uint8).pin_memory() for _ in range(2)]
    dbuf = [torch.empty(B, 3, S, S, dtype=torch.uint8, device=dev) for _ in range(2)]
    out = {}
    with torch.no_grad():
        for _ in range(3):
            bench(dbuf[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            bench(dbuf[0])
        torch.cuda.synchronize()
        out['resident_uint8_eager_img_s'] = round(B * steps / (time.perf_counter() - t0), 1)
        t0 = time.perf_counter()
        for i in range(steps):
            dbuf[0].copy_(host[i % 2], non_blocking=True)
            bench(dbuf[0])
        torch.cuda.synchronize()
        out['pcie_serial_img_s'] = round(B * steps / (time.perf_counter() - t0), 1)
        copy_stream = torch.cuda.Stream(dev)
        cur = torch.cuda.current_stream(dev)
        ready = [torch.cuda.Event(), torch.cuda.Event()]
        done = [torch.cuda.Event(), torch.cuda.Event()]
        with torch.cuda.stream(copy_stream):
            dbuf[0].copy_(host[0], non_blocking=True)
            ready[0].record(copy_stream)
        t0 = time.perf_counter()
        for i in range(steps):
            k, n = i % 2, (i + 1) % 2
            if i + 1 < steps:
                with torch.cuda.stream(copy_stream):
                    if i >= 1:
                        copy_stream.wait_event(done[n])          # the forward that last read dbuf[n] has finished
                    dbuf[n].copy_(host[n], non_blocking=True)
                    ready[n].record(copy_stream)
            cur.wait_event(ready[k])
            bench(dbuf[k])
            done[k].record(cur)
        torch.cuda.synchronize()
        out['pcie_overlapped_img_s'] = round(B * steps / (time.perf_counter() - t0), 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(10):
            dbuf[0].copy_(host[0], non_blocking=True)
        e1.record()
        e1.synchronize()
        out['h2d_GBps'] = round(10 * host[0].numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
    # the product helper: three batches in flight, one hipGraph per slot, resident bf16 input and host -> device uint8 input
    from ood_object_detection_amd.serving import PipelinedPredict
    pipe = PipelinedPredict(model, in_flight=3, graphs=True)
    xb = torch.randn(B, 3, S, S, device=dev).to(torch.bfloat16)

    def run(make_input, n):
        tickets = []
        for i in range(n):
            if len(tickets) >= 3:
                pipe.result(tickets.pop(0))
            tickets.append(pipe.submit(make_input(i)))
        for t in tickets:
            pipe.result(t)
    run(lambda i: xb, 6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(lambda i: xb, 30)
    torch.cuda.synchronize()
    out['pipelined_graphs_resident_bf16_img_s'] = round(B * 30 / (time.perf_counter() - t0), 1)
    pipe_u8 = PipelinedPredict(model, in_flight=3, graphs=True)
    stage = [torch.empty(B, 3, S, S, dtype=torch.uint8, device=dev) for _ in range(3)]

    def h2d(i):
        stage[i % 3].copy_(host[i % 2], non_blocking=True)
        return stage[i % 3]
    run_u8 = lambda n: None
    tickets = []
    for phase, n in (('warm', 6), ('timed', 30)):
        if phase == 'timed':
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        for i in range(n):
            if len(tickets) >= 3:
                pipe_u8.result(tickets.pop(0))
            tickets.append(pipe_u8.submit(h2d(i)))
        while tickets:
            pipe_u8.result(tickets.pop(0))
    torch.cuda.synchronize()
    out['pipelined_graphs_pcie_uint8_img_s'] = round(B * 30 / (time.perf_counter() - t0), 1)
    out['workload'] = 'tf_efficientdet_d0 640x640 batch 64 bf16, uint8 input from pinned host memory, eager launches'
    print(json.dumps(out))


if __name__ == '__main__':
    main()
