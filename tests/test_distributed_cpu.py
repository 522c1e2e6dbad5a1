"""world_size-2 checks of the multi-GPU plumbing on CPU (gloo): scene sharding, ordered result
collection, max-over-ranks timing and the packed loss all-reduce of ``SoftGroup.parse_losses``."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from softgroup_amd import dist as sgdist


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    sgdist.init_dist('gloo')
    try:
        n = 5
        mine = sgdist.shard_indices(n)
        results = [dict(scan=i, rank=rank) for i in mine]
        merged = sgdist.collect_results(results, n)
        tmax = sgdist.max_over_ranks(1.0 + rank)
        from softgroup_amd.model import SoftGroup
        model = SoftGroup(channels=16, num_blocks=2, semantic_only=True, semantic_classes=4)
        losses = dict(semantic_loss=torch.tensor(1.0 + rank), offset_loss=torch.tensor(3.0 * (rank + 1)))
        loss, log_vars = model.parse_losses(losses)
        ret[rank] = dict(mine=mine, merged=merged, tmax=tmax, loss=float(loss), log_vars=dict(log_vars))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_and_reduction():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]
    assert r0['mine'] == [0, 2, 4] and r1['mine'] == [1, 3, 0]          # padded by wrapping
    assert [m['scan'] for m in r0['merged']] == [0, 1, 2, 3, 4] and r1['merged'] is None
    assert r0['tmax'] == r1['tmax'] == 2.0
    # local loss stays local (it drives backward); logged values are rank means
    assert r0['loss'] == 4.0 and r1['loss'] == 8.0
    for r in (r0, r1):
        assert abs(r['log_vars']['semantic_loss'] - 1.5) < 1e-6
        assert abs(r['log_vars']['offset_loss'] - 4.5) < 1e-6
        assert abs(r['log_vars']['loss'] - 6.0) < 1e-6


def test_single_process_paths():
    assert sgdist.get_dist_info() == (0, 1)
    assert sgdist.shard_indices(3) == [0, 1, 2]
    assert sgdist.collect_results([1, 2, 3], 2) == [1, 2]
    assert sgdist.max_over_ranks(0.5) == 0.5


def _ddp_worker(rank, world, port, ret):
    """DistributedDataParallel over the TRAINABLE subset of the fine-tune configuration
    (softgroup_s3dis_fold5.yaml: fixed_modules = backbone + point-wise heads; reference tools/train.py:174
    wraps the whole model, DDP only reduces what requires grad): CPU stand-in with the real parameter
    names and shapes, gloo.  A comm hook counts the bytes that are all-reduced per step."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sgdist.init_dist('gloo')
    try:
        from softgroup_amd import synthetic
        from softgroup_amd.model import SoftGroup
        torch.manual_seed(0)
        real = SoftGroup(**synthetic.S3DIS_MODEL_CFG)        # BASELINE config 3 (softgroup_s3dis_fold5.yaml)
        named = [(n, p) for n, p in real.named_parameters() if p.requires_grad]

        class Heads(torch.nn.Module):            # same names (dots -> '/'), same shapes, used once each

            def __init__(self):
                super().__init__()
                self.p = torch.nn.ParameterDict({n.replace('.', '/'): torch.nn.Parameter(p.detach().clone())
                                                 for n, p in named})

            def forward(self, scale):
                return sum((v * scale).sum() for v in self.p.values())

        model = torch.nn.parallel.DistributedDataParallel(Heads())
        counted = []

        def hook(state, bucket):
            buf = bucket.buffer()
            counted.append(buf.numel() * buf.element_size())
            fut = dist.all_reduce(buf, async_op=True).get_future()
            return fut.then(lambda f: f.value()[0] / world)

        model.register_comm_hook(None, hook)
        loss = model(float(rank + 1))
        loss.backward()
        g = next(iter(model.module.p.values())).grad
        ret[rank] = dict(bytes=sum(counted), trainable=sum(p.numel() * p.element_size() for _, p in named),
                         names=[n for n, _ in named][:3] + [n for n, _ in named][-2:],
                         grad_mean=float(g.mean()), n_tensors=len(named))
    finally:
        dist.destroy_process_group()


def test_two_rank_ddp_allreduces_the_trainable_heads_only():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_ddp_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for r in (ret[0], ret[1]):
        # 2 919 208 B of gradients per step with the frozen backbone
        assert r['bytes'] == r['trainable'] == 2919208, r
        assert abs(r['grad_mean'] - 1.5) < 1e-6          # d/dp sum(p * s) = s, averaged over ranks 1 and 2
    assert ret[0]['names'][0].startswith('tiny_unet') and ret[0]['names'][-1].startswith('iou_score_linear')
