#!/usr/bin/env python
"""Where the pretrain step's time goes: the five stage functions of TrainEngine (backbone forward, BiFPN + heads forward,
loss, BiFPN + heads backward, backbone backward) each captured into its own hipGraph and replayed, ms per replay.

    python tools/train_sections.py [--batch 8] [--image 640] [--reps 20]

Under `rocprofv3 --kernel-trace --stats` with --only SECTION the kernel table of one stage results."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--image', type=int, default=640)
    ap.add_argument('--classes', type=int, default=90)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--model', default='tf_efficientdet_d0')
    ap.add_argument('--only', default='', help='replay only this section (bbf, fhf, loss, fhb, bbb)')
    args = ap.parse_args()
    from bench import build_model
    from ood_object_detection_amd.pretrain import set_bn_eval
    from ood_object_detection_amd.train_engine import TrainEngine
    from ood_object_detection_amd.effdet.loss import _DetectionLossFn
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    model = build_model(args.model, args.image, args.classes).to(dev).float()
    model.train()
    model.backbone.apply(set_bn_eval)
    eng = TrainEngine(model)
    B = args.batch
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(B, 3, args.image, args.image, device=dev, generator=g)
    st = {}

    def bbf():
        st['feats'], st['bbs'] = eng.bb_forward(x)

    def fhf():
        st['cls'], st['box'], st['fhs'] = eng.fh_forward(st['feats'])

    def loss():
        cls, box = st['cls'], st['box']
        N = cls.shape[1]
        if 'cls_t' not in st:
            st['cls_t'] = torch.randint(-2, args.classes, (B, N), device=dev)
            st['box_t'] = torch.randn(B, N, 4, device=dev)
            st['np'] = torch.full((B,), 30.0, device=dev)
        total, parts = _DetectionLossFn.apply(cls, box, st['cls_t'], st['box_t'], st['np'], 0.15, 0.1, 50.0, 0.0)
        st['g_cls'], st['g_box'] = torch.ones_like(cls) * 1e-3, torch.ones_like(box) * 1e-3

    def fhb():
        st['dfeats'], st['g1'] = eng.fh_backward(st['g_cls'], st['g_box'], st['fhs'])

    def bbb():
        st['g2'] = eng.bb_backward([d.contiguous() for d in st['dfeats']], st['bbs'])

    sections = [('bbf', bbf), ('fhf', fhf), ('loss', loss), ('fhb', fhb), ('bbb', bbb)]
    with torch.no_grad():
        for _ in range(2):
            for _, fn in sections:
                fn()
        torch.cuda.synchronize()
        total = 0.0
        for name, fn in sections:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                fn()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                fn()
            torch.cuda.synchronize()
            if args.only and args.only != name:
                gr.replay()
                torch.cuda.synchronize()
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            gr.replay()
            e0.record()
            for _ in range(args.reps):
                gr.replay()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.reps
            total += ms
            print('%-5s %8.3f ms' % (name, ms), flush=True)
        print('sum   %8.3f ms  (%d images: %.0f img/s without the optimizer)' % (total, B, 1e3 * B / max(total, 1e-9)))


if __name__ == '__main__':
    main()
