"""Per-layer-shape timing of the sparse-conv kernel INSIDE the native executor (sg_unet_forward): the
library brackets every conv call with HIP events (sg_spconv_profile) and hands back the per-call list
(sg_spconv_profile_detail).  Unlike tools/conv_layers.py (module path) this sees the executor's internal
row order.  Usage (GPU box): [SG_UNET_MORTON=0|1] python tools/conv_exec_layers.py [points] [reps]"""
import ctypes as C
import os
import sys
from collections import defaultdict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softgroup_amd import _lib as L, ops, synthetic  # noqa: E402
import softgroup_amd.spconv.pytorch as spconv  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    xyz, rgb, inst = synthetic.scene_s2(seed=1, n=n)
    b = synthetic.make_batch(xyz, rgb, instance_labels=inst)
    b = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    model = synthetic.build_model(seed=0)
    lib = L.lib()
    import contextlib
    ac = torch.autocast('cuda', dtype=torch.bfloat16) if os.environ.get('AUTOCAST') else contextlib.nullcontext()
    with torch.no_grad(), ac:
        vf = ops.voxelization(torch.cat((b['feats'], b['coords_float']), 1), b['p2v_map']).float()
        x = spconv.SparseConvTensor(vf, b['voxel_coords'].int(), b['spatial_shape'], 1)
        for _ in range(3):
            ref = model._unet_features(x)
        torch.cuda.synchronize()
        L.check(lib.sg_spconv_profile(1), 'sg_spconv_profile')
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * reps)]
        for r in range(reps):
            ev[2 * r].record()
            out = model._unet_features(x)
            ev[2 * r + 1].record()
        torch.cuda.synchronize()
    cap = 4096
    ms = np.zeros(cap, np.float32)
    dims = np.zeros((cap, 5), np.int32)
    calls = C.c_int(0)
    L.check(lib.sg_spconv_profile_detail(ms.ctypes.data, dims.ctypes.data, cap, C.byref(calls)), 'detail')
    L.check(lib.sg_spconv_profile(0), 'sg_spconv_profile')
    nc = min(calls.value, cap)
    agg = defaultdict(lambda: [0, 0.0])
    for i in range(nc):
        a = agg[tuple(dims[i, :4])]
        a[0] += 1
        a[1] += float(ms[i])
    # algorithmic gather/scatter bytes of a shape need the pair count P: not known here; the table is for A/B
    print(f'{"K":>3} {"Cin":>4} {"Cout":>4} {"M_out":>7} {"n/fwd":>6} {"us/launch":>9} {"ms/fwd":>8}')
    tot = 0.0
    for (m, k, ci, co), (cnt, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        tot += t / reps
        print(f'{k:>3} {ci:>4} {co:>4} {m:>7} {cnt / reps:>6.1f} {t / cnt * 1e3:>9.1f} {t / reps:>8.3f}')
    big = sum(t for (m, k, ci, co), (c, t) in agg.items() if m >= 20000) / reps
    fwd = sorted(ev[2 * r].elapsed_time(ev[2 * r + 1]) for r in range(reps))
    print(f'total conv ms/forward {tot:.3f} (levels >= 20k rows: {big:.3f}); calls/forward {nc / reps:.1f}; '
          f'forward incl. index build: median {fwd[len(fwd) // 2]:.3f} ms, min {fwd[0]:.3f}; '
          f'checksum {float(out.sum()):.6e} max|out-ref| {float((out - ref).abs().max()):.3e}; '
          f'SG_UNET_MORTON={os.environ.get("SG_UNET_MORTON", "default")}')


if __name__ == '__main__':
    main()
