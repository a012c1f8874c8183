"""Image sharding across the GPUs of one node (one process per GPU, no collective on the data path).

Inference over a batch is embarrassingly parallel (BN runs on running statistics), so rank r simply
takes the contiguous slice `shard_range(n, r, world)` of the images.  The only exchanges are optional:
`max_over_ranks` (benchmark timing) and `gather_detections`, which mirrors what the reference's evaluator
does with `all_gather_container` (effdet/distributed.py:255-278, effdet/evaluator.py:38-39) on the
fixed-shape `[B/G, max_det, 6]` + counts tensors (RCCL on GPUs, gloo on CPU).
"""
import torch


def shard_range(n, rank, world):
    """Contiguous, balanced slice [lo, hi) of n items for `rank` of `world`."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(value, device='cpu'):
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_detections(det, count):
    """All-gather per-rank detections [b_r, max_det, 6] / counts [b_r] (equal b_r on every rank)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return det, count
    world = dist.get_world_size()
    dets = [torch.empty_like(det) for _ in range(world)]
    counts = [torch.empty_like(count) for _ in range(world)]
    dist.all_gather(dets, det.contiguous())
    dist.all_gather(counts, count.contiguous())
    return torch.cat(dets, 0), torch.cat(counts, 0)


def allreduce_gradients(tensors, bucket_bytes=32 << 20, average=True):
    """Data-parallel gradient exchange for the pretrain step (pretrain.py:236-276 is single-GPU; DDP is this build's
    addition): gradients are packed into flat buckets and each bucket is all-reduced once
    (RCCL over xGMI on GPUs via the 'nccl' backend, gloo on CPU).  EfficientDet-D0 has ~3.9 M parameters
    (15.6 MB fp32): one 32 MB bucket = one collective per step, which is what a 7-link point-to-point xGMI
    fabric wants (few, large messages)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    world = dist.get_world_size()
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([t.reshape(-1) for t in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat /= world
        off = 0
        for t in bucket:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n
        bucket, size = [], 0

    for t in tensors:
        if t is None:
            continue
        nb = t.numel() * t.element_size()
        if bucket and (size + nb > bucket_bytes or t.dtype != bucket[0].dtype):
            flush()
        bucket.append(t)
        size += nb
    flush()


def rank_env(rank, world, port, base_env=None):
    """Environment of one rank started by `launch_ranks` (the variables torch.distributed's env:// rendezvous reads)."""
    import os
    env = dict(os.environ if base_env is None else base_env)
    env.update({'RANK': str(rank), 'LOCAL_RANK': str(rank), 'WORLD_SIZE': str(world), 'LOCAL_WORLD_SIZE': str(world),
                'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port)})
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # the host driver only supports dmabuf IPC (RCCL needs it)
    return env


def launch_ranks(world, argv, port=None, timeout=None):
    """Start `world` fresh processes `argv` (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), wait for all of
    them and return the worst exit code.  The caller must not have touched the GPU: the children are plain child processes
    (never an exec of this one), and this parent only waits.  If one rank fails the others are terminated (a rank that died
    before the rendezvous would leave the rest waiting for it)."""
    import socket
    import subprocess
    import time
    if world < 1:
        raise ValueError('world size must be >= 1')
    if port is None:
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
        s.close()
    procs = [subprocess.Popen(list(argv), env=rank_env(r, world, port)) for r in range(world)]
    t0, worst = time.time(), 0
    live = list(procs)
    while live:
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0:
                worst = rc if worst == 0 else worst
                for q in live:                                  # exact PIDs we started, nothing else
                    q.terminate()
        if timeout is not None and time.time() - t0 > timeout:
            for q in live:
                q.kill()
            worst = worst or 124
            break
        time.sleep(0.05)
    for p in procs:
        try:
            p.wait(timeout=30)
        except Exception:
            p.kill()
    return worst


def resolve_world(gpus_arg, environ=None):
    """(rank, local_rank, world, must_launch) for a `--gpus N` benchmark process.  WORLD_SIZE set (torchrun or `launch_ranks`
    started us): it must equal --gpus, else ValueError.  WORLD_SIZE unset and --gpus N > 1: this process is the parent that has
    to start N ranks itself (`must_launch`)."""
    import os
    env = os.environ if environ is None else environ
    if 'WORLD_SIZE' in env:
        world = int(env['WORLD_SIZE'])
        if world != int(gpus_arg):
            raise ValueError('--gpus %d but WORLD_SIZE=%d: start exactly --gpus ranks' % (gpus_arg, world))
        return int(env.get('RANK', '0')), int(env.get('LOCAL_RANK', '0')), world, False
    return 0, 0, int(gpus_arg), int(gpus_arg) > 1
