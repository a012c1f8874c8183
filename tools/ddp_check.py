#!/usr/bin/env python
"""Data-parallel correctness of PretrainStep on ONE box: N ranks (gloo here, so they can share a GPU; RCCL on a real node)
train on different batches; after every step all replicas must hold bit-identical weights, and the captured-graph path
(two graphs around the eager all-reduce) must give the same weights as the eager path.

    EFFDET_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 tools/ddp_check.py
"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch
import torch.distributed as dist


def run(graph, rank, world, dev, steps=5):
    from _models import seeded_model
    from ood_object_detection_amd.pretrain import PretrainStep
    size, B, C = 128, 2, 20
    model, cfg, nodes, sd = seeded_model('tf_efficientdet_d0', size, C, seed=23)
    model = model.to(dev).float()
    step = PretrainStep(model, graph=graph, graph_warmup=2)
    assert step.world == world
    g = torch.Generator().manual_seed(100 + rank)
    boxes = [torch.tensor([[10. + 3 * rank, 12., 70., 90.], [40., 30., 120., 100.]]), torch.tensor([[5., 5., 60. + 5 * rank, 50.]])]
    cls = [torch.tensor([3, 7]), torch.tensor([1 + rank])]
    target = {'bbox': [b.to(dev) for b in boxes], 'cls': [c.to(dev) for c in cls]}
    losses = []
    for _ in range(steps):
        x = torch.randint(0, 256, (B, 3, size, size), generator=g, dtype=torch.uint8).to(dev)
        losses.append(step(x, target)['loss'].item())
    flat = step.opt.flat_param.detach().cpu()
    digest = hashlib.sha256(flat.numpy().tobytes()).hexdigest()
    gathered = [None] * world
    dist.all_gather_object(gathered, digest)
    assert len(set(gathered)) == 1, 'replicas diverged: %s' % gathered
    return digest, losses


def main():
    rank, local_rank, world = int(os.environ['RANK']), int(os.environ['LOCAL_RANK']), int(os.environ['WORLD_SIZE'])
    backend = os.environ.get('EFFDET_DIST_BACKEND', 'nccl')
    dev = torch.device('cuda', local_rank if backend == 'nccl' else local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group(backend, rank=rank, world_size=world, **({'device_id': dev} if backend == 'nccl' else {}))
    d_eager, l_eager = run(False, rank, world, dev)
    d_graph, l_graph = run(True, rank, world, dev)
    assert d_eager == d_graph, 'graph path differs from eager path'
    assert l_eager == l_graph
    if rank == 0:
        print('DDP_CHECK_OK world=%d weights=%s losses=%s' % (world, d_eager[:16], ['%.3f' % v for v in l_eager]), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
