#!/usr/bin/env python
"""BASELINE config 5: pretrain.py's training iteration on tf_efficientdet_d0 at 640x640, float32, B images per GPU,
synthetic ground truth (M ~ U{1..20} boxes per image, min side 16 px, classes U{1..C}), labels assigned on
the GPU, DDP = one RCCL all-reduce of the flat gradient buffer per step.

    python tools/pretrain_bench.py --steps 10 --warmup 3
    python tools/pretrain_bench.py --gpus N            (starts N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/pretrain_bench.py --gpus N

Rank 0 prints one JSON line: steps/s, images/s (whole job), ms per step (max over ranks), all-reduce ms."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def synthetic_targets(B, size, C, seed, dev):
    g = torch.Generator().manual_seed(seed)
    boxes, cls = [], []
    for _ in range(B):
        m = int(torch.randint(1, 21, (1,), generator=g))
        y0 = torch.rand(m, generator=g) * (size - 16)
        x0 = torch.rand(m, generator=g) * (size - 16)
        h = 16 + torch.rand(m, generator=g) * (size - 16 - y0)
        w = 16 + torch.rand(m, generator=g) * (size - 16 - x0)
        boxes.append(torch.stack([y0, x0, y0 + h, x0 + w], 1).to(dev))
        cls.append(torch.randint(1, C + 1, (m,), generator=g).to(dev))
    return {'bbox': boxes, 'cls': cls}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--image', type=int, default=640)
    ap.add_argument('--classes', type=int, default=90)
    ap.add_argument('--model', default='tf_efficientdet_d0')
    ap.add_argument('--graph', action='store_true', help='replay the iteration from one captured hipGraph')
    args = ap.parse_args()
    # --gpus N > 1 without a launcher: this process is the parent, starts N rank processes of this command before any
    # torch.cuda call and exits with the worst child code; under a launcher WORLD_SIZE must equal --gpus (sharding.resolve_world)
    from ood_object_detection_amd.sharding import launch_ranks, resolve_world
    try:
        rank, local_rank, world, must_launch = resolve_world(args.gpus)
    except ValueError as e:
        raise SystemExit('pretrain_bench.py: %s' % e)
    if must_launch:
        worker = os.environ.get('EFFDET_BENCH_WORKER', os.path.abspath(__file__))
        sys.exit(launch_ranks(world, [sys.executable, worker] + sys.argv[1:]))
    if not torch.cuda.is_available():
        raise SystemExit('needs an MI355X: the product path has no CPU fallback')
    backend = os.environ.get('EFFDET_DIST_BACKEND', 'nccl')
    dev = torch.device('cuda', local_rank if backend == 'nccl' else local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend, rank=rank, world_size=world, **({'device_id': dev} if backend == 'nccl' else {}))
    from bench import build_model
    from ood_object_detection_amd.pretrain import PretrainStep
    model = build_model(args.model, args.image, args.classes).to(dev).float()
    with torch.no_grad():
        model.class_net.predict.conv_pw.bias.fill_(-4.59511985)          # reference init (-log(99)), efficientdet.py:513
    step = PretrainStep(model, graph=args.graph)
    B = args.batch
    x = torch.randint(0, 256, (B, 3, args.image, args.image), dtype=torch.uint8, device=dev,
                      generator=torch.Generator(device=dev).manual_seed(100 + rank))
    target = synthetic_targets(B, args.image, args.classes, 7 + rank, dev)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # Algorithmic HBM bytes of one step, counted (not modelled): every activation-sized tensor the training engine produces during
    # ONE eager step (forward activations kept for the backward, gradients, workspaces: train_engine._Ops.new) is written once and
    # read at least once -> 2 x their bytes; + the input batch, + parameters / Adam state (4 flat float32 buffers read, 3 written).
    from ood_object_detection_amd import train_engine as _te
    counted = {'bytes': 0, 'tensors': 0, 'on': False}
    _new = _te._Ops.new

    def counting_new(self, *shape):
        t = _new(self, *shape)
        if counted['on']:
            counted['bytes'] += t.numel() * 4
            counted['tensors'] += 1
        return t
    _te._Ops.new = counting_new
    losses = []
    for i in range(max(args.warmup, 4 if args.graph else 0)):
        counted['on'] = i == 1                                # the second step: tables recorded, still eager
        losses.append(step(x, target)['loss'].item())
    counted['on'] = False
    _te._Ops.new = _new
    barrier()
    ar = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(x, target, time_allreduce=world > 1)
        ar += step.last_allreduce_ms
    barrier()
    elapsed = time.perf_counter() - t0
    losses.append(out['loss'].item())
    ranks_seen, per_rank = 1, [round(B * args.steps / elapsed, 2)]
    if dist is not None:
        cdev = dev if backend == 'nccl' else 'cpu'
        rates = torch.zeros(world, device=cdev, dtype=torch.float64)
        rates[rank] = B * args.steps / elapsed
        t = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        ones = torch.ones(1, device=cdev, dtype=torch.float64)
        dist.all_reduce(ones)
        dist.all_reduce(rates)
        ranks_seen, per_rank = int(round(float(ones.item()))), [round(float(v), 2) for v in rates.tolist()]
    n_par = int(step.opt.flat_grad.numel())
    step_bytes = 2 * counted['bytes'] + int(x.numel()) + 7 * 4 * n_par
    ms = 1e3 * elapsed / args.steps
    # forward 7.79 GFLOP / image at 640 px, C = 90 (tools/roofline_table.py); dX and dW GEMMs repeat it: ~3x per training step
    flops = 3 * 7.79e9 * (args.image / 640.0) ** 2 * B if args.model == 'tf_efficientdet_d0' else None
    roofline = {'bound': 'hbm', 'achieved': round(step_bytes / (ms * 1e-3) / 1e9, 1), 'peak': 8000.0, 'unit': 'GB/s',
                'frac': round(step_bytes / (ms * 1e-3) / 1e9 / 8000.0, 4), 'traffic': None,
                'kernel': 'whole step (every launch of the captured iteration)',
                'algorithmic_bytes_per_step': int(step_bytes), 'activation_tensors_counted': counted['tensors'],
                'bytes_rule': '2 x bytes of every tensor train_engine._Ops.new produced in one eager step + input + 7 x 4 B x parameters',
                'hbm_floor_ms': round(step_bytes / 8e12 * 1e3, 3),
                'fp32_mfma_floor_ms': None if flops is None else round(flops / 157.3e12 * 1e3, 3)}
    if rank == 0:
        print(json.dumps({
            'metric': 'pretrain steps/sec, %s %dpx float32 (forward + loss + backward + grad all-reduce + clip + Adam)' % (args.model, args.image),
            'value': round(args.steps / elapsed, 3), 'unit': 'steps/sec', 'images_per_sec': round(world * B * args.steps / elapsed, 2),
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 2),
            'allreduce_ms_per_step': round(ar / max(args.steps, 1), 3), 'allreduce_bytes': int(step.opt.flat_grad.numel() * 4),
            'higher_is_better': True, 'scaling': 'weak', 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s %dx%d batch=%d/GPU C=%d pretrain step, labels assigned on the GPU' % (
                args.model, args.image, args.image, B, args.classes), 'global_batch': world * B,
                'parallelism': 'dp%d, one flat-gradient all-reduce per step' % world,
                'launch': 'hipgraph' if args.graph else 'eager'},
            'ranks': {'world_size': world, 'answered_all_reduce': ranks_seen, 'per_rank_images_per_sec': per_rank},
            'roofline': roofline,
            'loss_first_last': [round(losses[0], 4), round(losses[-1], 4)],
            'peak_mem_GB': round(torch.cuda.max_memory_allocated(dev) / 1e9, 2)}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
