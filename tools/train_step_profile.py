"""Full-model forward_train + backward + Adam, fp32, 8 steps: what `rocprofv3 --kernel-trace --stats` is pointed at
(tools/r06/r06_gpu_train_prof.sh).  Usage (GPU box): python tools/train_step_profile.py [points] [fp32|bf16]"""
import copy
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softgroup_amd import synthetic  # noqa: E402
from softgroup_amd.model import SoftGroup  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    bf16 = len(sys.argv) > 2 and sys.argv[2] == 'bf16'
    cfg = copy.deepcopy(synthetic.S3DIS_MODEL_CFG)
    cfg['test_cfg']['x4_split'] = False
    cfg['fixed_modules'] = []
    xyz, rgb, inst = synthetic.scene_s2(seed=21, n=n)
    batch = synthetic.make_batch(xyz, rgb, instance_labels=inst)
    batch['semantic_labels'] = batch['semantic_labels'].clamp(max=12)
    batch['instance_cls'] = batch['instance_cls'].clamp(max=12)
    torch.manual_seed(0)
    model = SoftGroup(**cfg).cuda()
    with torch.no_grad():
        model.semantic_linear[-1].weight.normal_(0, 20.0)
    model.train()
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4)
    ts = []
    for it in range(11):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=bf16):
            loss, _ = model(batch, return_loss=True)
        t1 = time.perf_counter()
        opt.zero_grad()
        loss.backward()
        t2 = time.perf_counter()
        opt.step()
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        ts.append([(b - a) * 1e3 for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t0, t4))])
    ts = ts[3:]
    med = [sorted(t[i] for t in ts)[len(ts) // 2] for i in range(5)]
    print('host ms: forward %.2f, backward %.2f, optimiser %.2f, final wait %.2f; step %.2f' % tuple(med), flush=True)


if __name__ == '__main__':
    main()
