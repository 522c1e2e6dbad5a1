"""One optimisation step of forward_train (BASELINE config 3 shape: S3DIS model section, frozen
backbone) in fp32 and under bf16 autocast: ms per step (forward + backward + Adam), median of 10.
Usage (GPU box): python tools/train_step_bench.py [points]"""
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
    for frozen in (True, False):
        cfg = copy.deepcopy(synthetic.S3DIS_MODEL_CFG)
        cfg['test_cfg']['x4_split'] = False
        if not frozen:
            cfg['fixed_modules'] = []
        xyz, rgb, inst = synthetic.scene_s2(seed=21, n=n)
        batch = synthetic.make_batch(xyz, rgb, instance_labels=inst)
        batch['semantic_labels'] = batch['semantic_labels'].clamp(max=12)
        batch['instance_cls'] = batch['instance_cls'].clamp(max=12)
        for name, ctx in (('fp32', torch.autocast('cuda', enabled=False)),
                          ('bf16 autocast', torch.autocast('cuda', dtype=torch.bfloat16))):
            # a fresh model and optimiser per precision: both lines start from the SAME weights and print the loss
            # of the same step (until round 5 the bf16 block continued the fp32 block's training: its loss was the
            # one after 26 steps, which read as a precision gap -- 12.5 vs 9.5 on the full model)
            torch.manual_seed(0)
            model = SoftGroup(**cfg).cuda()
            with torch.no_grad():
                model.semantic_linear[-1].weight.normal_(0, 20.0)
            model.train()
            opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4)
            ts = []
            for it in range(13):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                with ctx:
                    loss, _ = model(batch, return_loss=True)
                opt.zero_grad()
                loss.backward()
                opt.step()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            ts = sorted(ts[3:])
            print(f'{"frozen backbone" if frozen else "full model":>16} {name:>14}: {ts[len(ts) // 2]:8.2f} ms/step '
                  f'(loss {float(loss):.4f})', flush=True)


if __name__ == '__main__':
    main()
