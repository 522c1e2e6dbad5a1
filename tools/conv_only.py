"""Only the backbone U-Net of the bench scene (index build + 79 sparse-conv launches per forward),
for profiling runs that would otherwise spend their time in the rest of the scan.
Usage (GPU box): python tools/conv_only.py [forwards] [points]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softgroup_amd import ops, synthetic  # noqa: E402
import softgroup_amd.spconv.pytorch as spconv  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 150000
    xyz, rgb, inst = synthetic.scene_s2(seed=1, n=n)
    b = synthetic.make_batch(xyz, rgb, instance_labels=inst)
    b = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    model = synthetic.build_model(seed=0)
    with torch.no_grad():
        vf = ops.voxelization(torch.cat((b['feats'], b['coords_float']), 1), b['p2v_map'])
        x = spconv.SparseConvTensor(vf, b['voxel_coords'].int(), b['spatial_shape'], 1)
        model._unet_features(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = model._unet_features(x)
        torch.cuda.synchronize()
    print(f'{reps} backbone forwards, {(time.perf_counter() - t0) / reps * 1e3:.3f} ms each, checksum {float(out.sum()):.6e}')


if __name__ == '__main__':
    main()
