"""GPU-free stand-in for a bench.py rank (tests/test_multiproc.py): takes part in the rendezvous `bench.py --gpus N` sets up for its
children, all-reduces over gloo and lets rank 0 print the bench line's rank bookkeeping."""
import argparse
import json
import os
import sys

import torch
import torch.distributed as dist

ap = argparse.ArgumentParser()
ap.add_argument('--gpus', type=int, default=1)
ap.add_argument('--steps', type=int, default=1)
ap.add_argument('--warmup', type=int, default=0)
args, _ = ap.parse_known_args()
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
assert world == args.gpus and int(os.environ['LOCAL_RANK']) == rank and os.environ['MASTER_ADDR'] == '127.0.0.1'
if rank == int(os.environ.get('EFFDET_FAKE_FAIL_RANK', '-1')):
    sys.exit(7)                                      # dies before the rendezvous: the launcher must stop the others
dist.init_process_group('gloo', rank=rank, world_size=world)
ones = torch.ones(1, dtype=torch.float64)
dist.all_reduce(ones)
rates = torch.zeros(world, dtype=torch.float64)
rates[rank] = 100.0 + rank
dist.all_reduce(rates)
if rank == 0:
    print(json.dumps({'n_gpus': world, 'ranks': {'world_size': dist.get_world_size(), 'answered_all_reduce': int(ones.item()),
                                                 'per_rank_images_per_sec': rates.tolist()}}), flush=True)
dist.barrier()
dist.destroy_process_group()
