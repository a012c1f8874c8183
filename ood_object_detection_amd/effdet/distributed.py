"""`effdet.distributed` helpers the fork's scripts import (infer.py:14, pretrain.py:12; reference: effdet/distributed.py:255-278).

Inference shards images over the GPUs of a node with no collective on the data path; gathering the per-rank
detections / metrics for logging is the one optional exchange, over `torch.distributed` (backend 'nccl' = RCCL over xGMI on
ROCm, 'gloo' in the CPU tests)."""
import torch
import torch.distributed as dist


def all_gather_container(container, group=None, cat_dim=0):
    """all_gather of a tensor, or of every tensor of a list / tuple / dict, concatenated along cat_dim (equal shapes per rank)."""
    group = group or dist.group.WORLD
    world = dist.get_world_size(group)

    def gather(t):
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t, group=group)
        return torch.cat(parts, dim=cat_dim)

    if isinstance(container, dict):
        return {k: gather(v) for k, v in container.items()}
    if isinstance(container, (list, tuple)):
        out = [gather(v) for v in container]
        return tuple(out) if isinstance(container, tuple) else out
    if not isinstance(container, torch.Tensor):
        raise TypeError('expected a tensor or a dict / list / tuple of tensors')
    return gather(container)


def reduce_dict(input_dict, average=True):
    """sum (rank 0: optionally the mean) of every scalar tensor of a dict over the ranks; no-op for a single process"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() < 2:
        return input_dict
    with torch.no_grad():
        names = sorted(input_dict.keys())
        values = torch.stack([input_dict[k] for k in names], dim=0)
        dist.reduce(values, dst=0)
        if dist.get_rank() == 0 and average:
            values /= dist.get_world_size()
        return dict(zip(names, values))
