"""Developer tool: per-tensor differences between the native training executor and the module path.
Usage (GPU box): python tools/train_exec_diag.py"""
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import softgroup_amd.spconv.pytorch as spconv  # noqa: E402
from softgroup_amd.spconv.unet_train import UNetTrainExecutor  # noqa: E402
from test_unet_train_gpu import Net, _randomise, _voxels  # noqa: E402


def rel(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-6)


def run(planes, cin, n, shape, batch):
    rng = np.random.default_rng(len(planes) + n)
    idx = _voxels(rng, n, shape, batch)
    M = idx.shape[0]
    torch.manual_seed(3)
    ref = Net(planes, cin).cuda().train()
    _randomise(ref, 5)
    net = copy.deepcopy(ref)
    x0 = torch.randn(M, cin if cin is not None else planes[0], device='cuda')
    g_out = torch.randn(M, planes[0], device='cuda')
    xr = x0.clone().requires_grad_(True)
    out_r = ref(spconv.SparseConvTensor(xr, idx, shape, batch))
    out_r.backward(g_out)
    ex = UNetTrainExecutor(net.unet, net.input_conv, net.output_layer)
    xe = x0.clone().requires_grad_(True)
    out_e = ex(spconv.SparseConvTensor(xe, idx, shape, batch))
    out_e.backward(g_out)
    def l2(a, b):
        return float((a.double() - b.double()).norm()) / max(float(b.double().norm()), 1e-12)

    rows = [(k, rel(pe.grad, pr.grad), l2(pe.grad, pr.grad))
            for (k, pe), (_, pr) in zip(net.named_parameters(), ref.named_parameters())]
    wm = max(rows, key=lambda r: r[1])
    wl = max(rows, key=lambda r: r[2])
    print(f'== planes {planes} cin {cin} voxels {M}: out {rel(out_e.detach(), out_r.detach()):.1e} '
          f'g_in max {rel(xe.grad, xr.grad):.1e} l2 {l2(xe.grad, xr.grad):.1e}; parameters: '
          f'{sum(r[1] > 1e-4 for r in rows)} of {len(rows)} above 1e-4 in max norm, worst max {wm[1]:.1e} ({wm[0]}), '
          f'worst l2 {wl[2]:.1e} ({wl[0]})')


if __name__ == '__main__':
    run([16, 32, 48], 6, 60000, [128, 96, 48], 2)
    run([16, 32, 48], None, 60000, [128, 96, 48], 2)
    run([16, 32], None, 60000, [128, 96, 48], 2)
    run([16], 6, 60000, [128, 96, 48], 2)
    run([32, 64, 96], 6, 60000, [128, 96, 48], 2)
    run([32, 64, 96], None, 60000, [128, 96, 48], 2)
