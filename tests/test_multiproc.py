"""N > 1 path on CPU: two gloo ranks shard a batch, gather fixed-shape detections, reduce the bench timing."""
import os
import socket

import pytest

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ood_object_detection_amd.sharding import allreduce_gradients, gather_detections, max_over_ranks, shard_range


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    n = 8
    lo, hi = shard_range(n, rank, world)
    full = torch.arange(n * 100 * 6, dtype=torch.float32).reshape(n, 100, 6)
    counts = torch.arange(n, dtype=torch.int32)
    det, cnt = gather_detections(full[lo:hi], counts[lo:hi])
    t = max_over_ranks(1.0 + rank)
    # gradient exchange: three tensors, bucket size small enough to force two buckets
    grads = [torch.full((5, 3), float(rank + 1)), torch.arange(7, dtype=torch.float32) * (rank + 1), torch.ones(2) * rank]
    allreduce_gradients(grads, bucket_bytes=64)
    g_ok = bool(torch.allclose(grads[0], torch.full((5, 3), 1.5)) and torch.allclose(grads[1], torch.arange(7.) * 1.5)
                and torch.allclose(grads[2], torch.ones(2) * 0.5))
    # effdet.distributed.all_gather_container (reference: effdet/distributed.py:255-278; imported by infer.py:14, pretrain.py:12)
    from ood_object_detection_amd.effdet.distributed import all_gather_container, reduce_dict
    a = torch.full((2, 3), float(rank))
    got_t = all_gather_container(a)
    got_d = all_gather_container({'x': a, 'y': torch.tensor([rank])})
    got_l = all_gather_container((a, a + 1), cat_dim=1)
    want = torch.cat([torch.full((2, 3), float(r)) for r in range(world)], 0)
    c_ok = bool(torch.equal(got_t, want) and torch.equal(got_d['x'], want) and got_d['y'].tolist() == list(range(world)) and
                isinstance(got_l, tuple) and got_l[1].shape == (2, 3 * world))
    red = reduce_dict({'b': torch.tensor(float(rank + 1)), 'a': torch.tensor(2.0)})
    if rank == 0:
        c_ok = c_ok and abs(float(red['b']) - 1.5) < 1e-6 and abs(float(red['a']) - 2.0) < 1e-6
    q.put((rank, lo, hi, bool(torch.equal(det, full)), bool(torch.equal(cnt, counts)), t, g_ok and c_ok))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    for n in (1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_ranks_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [(o[1], o[2]) for o in out] == [(0, 4), (4, 8)]
    assert all(o[3] and o[4] for o in out)
    assert all(o[5] == 2.0 for o in out)           # max over ranks of (1.0, 2.0)
    assert all(o[6] for o in out)                  # bucketed gradient all-reduce (mean)


def _run_bench_parent(extra_args, env_extra=None, script='bench.py'):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env['EFFDET_BENCH_WORKER'] = os.path.join(root, 'tests', '_fake_bench_worker.py')
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, script)] + extra_args, env=env, capture_output=True, text=True, timeout=300)


@pytest.mark.parametrize('n', [2, 8])
@pytest.mark.parametrize('script', ['bench.py', os.path.join('tools', 'pretrain_bench.py')])
def test_bench_gpus_n_launches_n_ranks_itself(script, n):
    """`python bench.py --gpus N` with no launcher around it: the parent starts N rank processes (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set) BEFORE touching the GPU - here GPU-free workers over gloo, two and eight of them (the driver's largest node) -
    waits, and exits with the worst child code"""
    import json
    r = _run_bench_parent(['--gpus', str(n), '--steps', '1', '--warmup', '0'], script=script)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == n and line['ranks']['answered_all_reduce'] == n and line['ranks']['world_size'] == n
    assert line['ranks']['per_rank_images_per_sec'] == [100.0 + i for i in range(n)]
    # a rank that dies takes the job down with its exit code (the other ranks would wait for it in the rendezvous for ever)
    r = _run_bench_parent(['--gpus', str(n)], env_extra={'EFFDET_FAKE_FAIL_RANK': str(n - 1)}, script=script)
    assert r.returncode == 7
    # started by a launcher with the wrong number of ranks: refuse
    r = _run_bench_parent(['--gpus', str(n)], env_extra={'WORLD_SIZE': str(n + 1), 'RANK': '0', 'LOCAL_RANK': '0'}, script=script)
    assert r.returncode != 0 and 'WORLD_SIZE=%d' % (n + 1) in (r.stderr + r.stdout)


def test_resolve_world():
    from ood_object_detection_amd.sharding import resolve_world
    assert resolve_world(1, {}) == (0, 0, 1, False)
    assert resolve_world(4, {}) == (0, 0, 4, True)
    assert resolve_world(4, {'WORLD_SIZE': '4', 'RANK': '3', 'LOCAL_RANK': '3'}) == (3, 3, 4, False)
    with pytest.raises(ValueError):
        resolve_world(8, {'WORLD_SIZE': '1'})
