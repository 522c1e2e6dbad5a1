"""Developer tool: is a scan bound by the host (kernel-launch enqueue) or by the GPU?
Runs K pipelined scans and reports (a) the wall time until the last model() call RETURNED (host
enqueue time; results not yet resolved), (b) the wall time until everything finished.
Usage (GPU box): python tools/host_enqueue.py [points]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softgroup_amd import synthetic  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
    xyz, rgb, inst = synthetic.scene_s2(seed=1, n=n)
    batch = synthetic.make_batch(xyz, rgb, instance_labels=inst)
    batch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    model = synthetic.build_model(seed=0)
    K = 20
    with torch.no_grad():
        for _ in range(3):
            model(batch).resolve()
        for mode in ('pipelined', 'unpipelined'):
            model.async_results = mode == 'pipelined'
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rets = [model(batch) for _ in range(K)]
            t_host = time.perf_counter() - t0
            for r in rets:
                r.resolve() if hasattr(r, 'resolve') else None
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
            print(f'{mode}: host enqueue {t_host / K * 1e3:.2f} ms/scan, everything done {t_all / K * 1e3:.2f} ms/scan')
        # the backbone alone: host time of the native executor call vs its GPU time
        from softgroup_amd import ops
        import softgroup_amd.spconv.pytorch as spconv
        b = batch
        feats = torch.cat((b['feats'], b['coords_float']), 1)
        vf = ops.voxelization(feats, b['p2v_map'])
        x = spconv.SparseConvTensor(vf, b['voxel_coords'].int(), b['spatial_shape'], 1)
        for _ in range(3):
            model._unet_features(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(K):
            model._unet_features(x)
        e1.record()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        print(f'backbone U-Net only: host enqueue {t_host / K * 1e3:.2f} ms, GPU {e0.elapsed_time(e1) / K:.2f} ms per call')


if __name__ == '__main__':
    main()
