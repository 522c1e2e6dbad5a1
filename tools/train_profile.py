"""Developer tool: where does one training step of BASELINE config 3's shape go (4 scenes x 150 k
points, frozen backbone, forward_train + backward + Adam)?  (1) wall time per stage with a device
synchronisation on either side, (2) cProfile of the unsynchronised step.
Usage (GPU box): python tools/train_profile.py [steps] [stages|host|plain][16]   (suffix 16: under bf16 autocast)"""
import cProfile
import copy
import io
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softgroup_amd import synthetic  # noqa: E402
from softgroup_amd.data import collate_device, make_item  # noqa: E402
from softgroup_amd.model import SoftGroup  # noqa: E402


def setup(points=150000, scenes=4):
    cfg = copy.deepcopy(synthetic.S3DIS_MODEL_CFG)
    torch.manual_seed(0)
    model = SoftGroup(**cfg).cuda()
    with torch.no_grad():
        model.semantic_linear[-1].weight.normal_(0, 20.0)
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-4)
    items = []
    for i in range(scenes):
        x, c, ins = synthetic.scene_s2(seed=31 + i, n=points)
        sem = np.where(ins >= 0, 2 + ins % 11, 0).astype(np.int64)
        items.append(make_item(x, c, 50, sem, ins, f'crop_{i}'))
    batch = collate_device(items)
    batch['instance_cls'] = batch['instance_cls'].clamp(min=0)
    return model, opt, batch


AUTOCAST = False


def step(model, opt, batch):
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=AUTOCAST):
        loss, _ = model(batch, return_loss=True)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    global AUTOCAST
    mode = sys.argv[2] if len(sys.argv) > 2 else 'stages'
    if mode.endswith('16'):
        AUTOCAST, mode = True, mode[:-2]
    model, opt, batch = setup()
    for _ in range(3):
        step(model, opt, batch)
    torch.cuda.synchronize()
    if mode == 'plain':
        t0 = time.perf_counter()
        for _ in range(n):
            step(model, opt, batch)
        torch.cuda.synchronize()
        print(f'{(time.perf_counter() - t0) / n * 1e3:.3f} ms/step')
        return
    if mode == 'host':
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        pr.enable()
        for _ in range(n):
            step(model, opt, batch)
        pr.disable()
        torch.cuda.synchronize()
        print(f'{(time.perf_counter() - t0) / n * 1e3:.3f} ms/step under the profiler')
        for key in ('cumulative', 'tottime'):
            s = io.StringIO()
            pstats.Stats(pr, stream=s).sort_stats(key).print_stats(60)
            print(s.getvalue()[:12000])
        return
    acc = {}

    def timed(obj, name, label=None):
        fn = getattr(obj, name)

        def wrap(*a, **k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn(*a, **k)
            torch.cuda.synchronize()
            acc[label or name] = acc.get(label or name, 0.0) + (time.perf_counter() - t0) * 1e3
            return r
        setattr(obj, name, wrap)

    for name in ('forward_backbone', 'point_wise_loss', 'forward_grouping', 'clusters_voxelization',
                 'forward_instance', 'instance_loss', 'parse_losses'):
        timed(model, name)
    tot = bw = op = 0.0
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=AUTOCAST):
            loss, _ = model(batch, return_loss=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        opt.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        opt.step()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        tot += (t3 - t0) * 1e3
        bw += (t2 - t1) * 1e3
        op += (t3 - t2) * 1e3
    for k, v in acc.items():
        print(f'{k:>24}: {v / n:8.3f} ms')
    print(f'{"backward":>24}: {bw / n:8.3f} ms')
    print(f'{"optimizer":>24}: {op / n:8.3f} ms')
    print(f'{"step (synchronised)":>24}: {tot / n:8.3f} ms')


if __name__ == '__main__':
    main()
