"""The differentiable oracle stand-ins behind the GRADIENT goldens (oracle/facade.py: _GatherConvFn,
_VoxelizationFn, _GlobalAvgPoolFn; tests/golden/make_ref_train.py) pinned on the CPU: the sparse
convolutions' gradients against torch autograd through the DENSE formulation that defines their
semantics (F.conv3d / F.conv_transpose3d on the densified tensor, the same equivalence
tests/golden/make_golden.py uses for the forward values; SURVEY 2.4 -- spconv itself is not vendored),
the voxel pooling and ROI pooling gradients against autograd through plain torch indexing."""
import numpy as np
import torch
import torch.nn.functional as F

import oracle
from oracle import facade


def setup_module(module):
    oracle.build()


def _sparse(seed, D=8, B=2, cin=6, p=0.35):
    rng = np.random.default_rng(seed)
    occ = rng.random((B, D, D, D)) < p
    idx = np.argwhere(occ).astype(np.int32)
    idx = idx[rng.permutation(len(idx))]
    f = rng.standard_normal((len(idx), cin)).astype(np.float32)
    return idx, f


def _dense(idx, feats, D, B):
    d = torch.zeros(B, feats.shape[1], D, D, D, dtype=feats.dtype)
    i = torch.from_numpy(idx.astype(np.int64))
    return d.index_put((i[:, 0], slice(None), i[:, 1], i[:, 2], i[:, 3]), feats) if False else \
        d.permute(0, 2, 3, 4, 1).index_put((i[:, 0], i[:, 1], i[:, 2], i[:, 3]), feats).permute(0, 4, 1, 2, 3)


def _at(dense, idx):
    i = torch.from_numpy(idx.astype(np.int64))
    return dense.permute(0, 2, 3, 4, 1)[i[:, 0], i[:, 1], i[:, 2], i[:, 3]]


def _close(a, b, tol=2e-5):
    scale = max(float(b.abs().max()), 1e-6)
    assert float((a.double() - b.double()).abs().max()) <= tol * scale, (float((a.double() - b.double()).abs().max()), scale)


def test_subm_conv_gradients_equal_dense_autograd():
    D, B, cin, cout = 8, 2, 6, 10
    idx, f = _sparse(1, D, B, cin)
    g = torch.Generator().manual_seed(2)
    conv = facade.SubMConv3d(cin, cout, 3, padding=1, bias=False, indice_key='k')
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.2)
    x = torch.from_numpy(f).requires_grad_(True)
    out = conv(facade.SparseConvTensor(x, torch.from_numpy(idx), [D] * 3, B)).features
    go = torch.randn(out.shape, generator=g)
    out.backward(go)
    # dense: conv3d over the densified input, read at the active sites, same upstream gradient
    xd = torch.from_numpy(f).double().requires_grad_(True)
    wd = conv.weight.detach().double().requires_grad_(True)
    od = _at(F.conv3d(_dense(idx, xd, D, B), wd.permute(0, 4, 1, 2, 3), padding=1), idx)
    _close(out.detach(), od.detach())
    od.backward(go.double())
    _close(x.grad, xd.grad)
    _close(conv.weight.grad, wd.grad)


def test_strided_and_inverse_conv_gradients_equal_dense_autograd():
    D, B, cin, cmid = 8, 2, 6, 8
    idx, f = _sparse(3, D, B, cin)
    g = torch.Generator().manual_seed(4)
    down = facade.SparseConv3d(cin, cmid, 2, stride=2, bias=False, indice_key='d')
    up = facade.SparseInverseConv3d(cmid, cin, 2, bias=False, indice_key='d')
    with torch.no_grad():
        down.weight.copy_(torch.randn(down.weight.shape, generator=g) * 0.3)
        up.weight.copy_(torch.randn(up.weight.shape, generator=g) * 0.3)
    x = torch.from_numpy(f).requires_grad_(True)
    y = down(facade.SparseConvTensor(x, torch.from_numpy(idx), [D] * 3, B))
    z = up(y).features
    go = torch.randn(z.shape, generator=g)
    z.backward(go)
    xd = torch.from_numpy(f).double().requires_grad_(True)
    w1 = down.weight.detach().double().requires_grad_(True)
    w2 = up.weight.detach().double().requires_grad_(True)
    yd = F.conv3d(_dense(idx, xd, D, B), w1.permute(0, 4, 1, 2, 3), stride=2)
    # coarse sites without an active child are not sites of the sparse tensor: their dense value is 0
    # anyway (no bias), so the transposed conv sees the same input
    zd = _at(F.conv_transpose3d(yd, w2.permute(4, 0, 1, 2, 3), stride=2), idx)
    _close(z.detach(), zd.detach())
    zd.backward(go.double())
    _close(x.grad, xd.grad)
    _close(down.weight.grad, w1.grad)
    _close(up.weight.grad, w2.grad)


def test_voxelization_and_roi_pool_gradients():
    ops = facade._ops_module()
    rng = np.random.default_rng(5)
    n, c = 500, 7
    coords = np.concatenate([np.zeros((n, 1), np.int64), rng.integers(0, 6, (n, 3))], 1)
    _, p2v, v2p = ops.voxelization_idx(torch.from_numpy(coords), 1)
    feats = torch.randn(n, c, generator=torch.Generator().manual_seed(6), requires_grad=True)
    vox = ops.voxelization(feats, v2p)                      # mean of the points of a voxel (mode 4)
    go = torch.randn(vox.shape, generator=torch.Generator().manual_seed(7))
    vox.backward(go)
    fd = feats.detach().double().requires_grad_(True)
    cnt = torch.bincount(p2v.long(), minlength=vox.shape[0]).double()
    ref = torch.zeros(vox.shape[0], c, dtype=torch.double).index_add(0, p2v.long(), fd) / cnt[:, None]
    _close(vox.detach(), ref.detach())
    ref.backward(go.double())
    _close(feats.grad, fd.grad)
    # ROI average pool over proposal segments
    offs = torch.tensor([0, 40, 41, 200, 500], dtype=torch.int32)
    f2 = torch.randn(n, c, generator=torch.Generator().manual_seed(8), requires_grad=True)
    pooled = ops.global_avg_pool(f2, offs)
    go2 = torch.randn(pooled.shape, generator=torch.Generator().manual_seed(9))
    pooled.backward(go2)
    f2d = f2.detach().double().requires_grad_(True)
    ref2 = torch.stack([f2d[int(offs[i]):int(offs[i + 1])].mean(0) for i in range(4)])
    _close(pooled.detach(), ref2.detach())
    ref2.backward(go2.double())
    _close(f2.grad, f2d.grad)
