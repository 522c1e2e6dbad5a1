"""Developer tool: how many kernel offsets the mask-sorted 32-row tiles of each U-Net level hold
(popcount of the tile masks of the SubM plans of the bench scene) -- the number of (offset, slice)
items a conv unit splits over its waves is popcount * Cin / 32.
Usage (GPU box): python tools/tile_offsets.py [points]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softgroup_amd import ops, synthetic  # noqa: E402
import softgroup_amd.spconv.pytorch as spconv  # noqa: E402
from softgroup_amd.spconv import core  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
    xyz, rgb, inst = synthetic.scene_s2(seed=1, n=n)
    b = synthetic.make_batch(xyz, rgb, instance_labels=inst)
    b = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    idx = b['voxel_coords'].int()
    shape = list(b['spatial_shape'])
    print(f'{"level":>5} {"rows":>7} {"tiles":>6} {"mean":>5} {"p50":>4} {"p90":>4} {"max":>4}  '
          f'share of tiles with <=2 / 3-4 / 5-8 / 9-16 / >16 offsets   share of (tile, offset) pairs in tiles > 8')
    for level in range(7):
        rule = core.SubMRule(idx, shape)
        m = rule.plan.tile_mask.cpu().numpy().astype(np.uint32)[:(rule.plan.num_out + 31) // 32]
        pc = np.array([bin(int(v)).count('1') for v in m])
        edges = [(0, 2), (3, 4), (5, 8), (9, 16), (17, 27)]
        share = [float(((pc >= lo) & (pc <= hi)).mean()) for lo, hi in edges]
        heavy = float(pc[pc > 8].sum() / max(pc.sum(), 1))
        print(f'{level:5d} {idx.shape[0]:7d} {len(pc):6d} {pc.mean():5.1f} {int(np.median(pc)):4d} '
              f'{int(np.percentile(pc, 90)):4d} {pc.max():4d}  ' + ' / '.join(f'{s:.2f}' for s in share)
              + f'   {heavy:.2f}')
        down = core.DownRule(idx, shape, 1)
        idx, shape = down.out_indices, down.out_spatial_shape
        if idx.shape[0] == 0:
            break


if __name__ == '__main__':
    main()
