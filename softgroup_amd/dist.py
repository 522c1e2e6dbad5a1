"""Multi-GPU plumbing of the hot path: scenes shard one per rank, no data-path collective.

Mirrors what the reference does around its test loop: ``init_dist`` (softgroup/util/dist.py:27-31),
``DistributedSampler`` round-robin sharding (softgroup/data/__init__.py:31) and the result
collection of ``collect_results_cpu`` (softgroup/util/dist.py:76-112: rank-interleaved merge,
padding truncated) -- here through ``all_gather_object`` instead of pickles on a shared tmp dir.
Backend 'nccl' is RCCL on ROCm (xGMI inside a node); 'gloo' is used by the CPU tests."""
import os

import torch
import torch.distributed as dist


def get_dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_dist(backend='nccl'):
    """env:// rendezvous as launched by torchrun; one process per GPU."""
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    rank = int(os.environ['RANK'])
    if backend == 'nccl':
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', rank % max(torch.cuda.device_count(), 1))))
    dist.init_process_group(backend=backend)
    return get_dist_info()


def shard_indices(n_items, rank=None, world=None):
    """Indices of the scenes rank `rank` processes: i = rank, rank+world, ... padded by wrapping so
    that every rank gets ceil(n/world) items (DistributedSampler semantics, shuffle off)."""
    if rank is None:
        rank, world = get_dist_info()
    per = (n_items + world - 1) // world
    idx = list(range(n_items))
    idx += idx[:per * world - n_items]
    return idx[rank:per * world:world]


def collect_results(result_part, size):
    """Gather per-rank result lists to rank 0 in dataset order (None elsewhere)."""
    rank, world = get_dist_info()
    if world == 1:
        return result_part[:size]
    parts = [None] * world
    dist.all_gather_object(parts, result_part)
    if rank != 0:
        return None
    ordered = []
    for group in zip(*parts):          # rank-interleaved: item i came from rank i % world
        ordered.extend(group)
    return ordered[:size]


def max_over_ranks(value, device=None):
    """max of a python float over ranks (bench timing contract)"""
    rank, world = get_dist_info()
    if world == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
