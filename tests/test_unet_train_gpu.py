"""The native TRAINING executor of the sparse U-Net (csrc/unet_train.hip: sg_unet_train_forward /
sg_unet_train_backward) against the module path it replaces -- the modules of model/blocks.py in
train() mode under torch autograd (reference softgroup/model/blocks.py:44-143 trained through
spconv's autograd functions and torch.nn.BatchNorm1d, tools/train.py:44-62), which
tests/test_train_gpu.py pins to the reference's own forward_train.  Same weights, same input:
the output and the BatchNorm running statistics must agree within fp32 summation-order noise
(north-star tolerance 1e-4 of the tensor's scale), the gradients in relative L2 (see _close_l2)."""
import copy
import functools

import numpy as np
import pytest
import torch
from torch import nn

import softgroup_amd.spconv.pytorch as spconv
from softgroup_amd.model.blocks import ResidualBlock, UBlock
from softgroup_amd.spconv.unet_train import UNetTrainExecutor

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _voxels(rng, n, extent, batch):
    pts = rng.random((n, 3)) * extent
    pts[:, 2] = (np.sin(pts[:, 0] * 0.3) + np.cos(pts[:, 1] * 0.2)) * 3 + extent[2] / 2 + rng.normal(0, 0.6, n)
    v = np.clip(np.floor(pts), 0, np.array(extent) - 1).astype(np.int64)
    b = np.sort(rng.integers(0, batch, n))
    key = ((b * extent[0] + v[:, 0]) * extent[1] + v[:, 1]) * extent[2] + v[:, 2]
    _, first = np.unique(key, return_index=True)
    first = np.sort(first)
    return torch.from_numpy(np.concatenate([b[first, None], v[first]], 1).astype(np.int32)).to(DEV)


class Net(nn.Module):

    def __init__(self, planes, cin=None, reps=2):
        super().__init__()
        norm_fn = functools.partial(nn.BatchNorm1d, eps=1e-4, momentum=0.1)
        self.input_conv = None
        if cin is not None:
            self.input_conv = spconv.SparseSequential(
                spconv.SubMConv3d(cin, planes[0], kernel_size=3, padding=1, bias=False, indice_key='subm1'))
        self.unet = UBlock(planes, norm_fn, reps, ResidualBlock, indice_key_id=1)
        self.output_layer = spconv.SparseSequential(norm_fn(planes[0]), nn.ReLU())

    def forward(self, x):
        if self.input_conv is not None:
            x = self.input_conv(x)
        return self.output_layer(self.unet(x)).features


def _randomise(net, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, nn.BatchNorm1d):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.2)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)


def _close(a, b, what, tol=1e-4):
    scale = max(float(b.abs().max()), 1e-6)
    err = float((a - b).abs().max()) / scale
    assert err <= tol, f'{what}: max |diff| / scale = {err:.2e}'
    return err


def _close_l2(a, b, what, tol=1e-2):
    """gradients: relative L2 distance.  Two fp32 implementations of the forward differ by ~1e-6, so
    a handful of the ~10^6 pre-activations of a layer that lie within 1e-6 of zero get the opposite
    ReLU mask; each such flip moves one term of a gradient sum by O(1) -- 1e-3..1e-2 of the largest
    entry of a small parameter gradient (measured: the cases without a flip agree to < 1e-4 in the
    maximum norm, tools/train_exec_diag.py), but next to nothing in L2.  A wrong table, weight layout
    or BatchNorm formula is an O(1) relative error."""
    err = float((a.double() - b.double()).norm()) / max(float(b.double().norm()), 1e-12)
    assert err <= tol, f'{what}: |diff|_2 / |ref|_2 = {err:.2e}'
    return err


CASES = [
    ('tiny_unet', [32, 64], None, 9000, [20, 20, 20], 40),          # the refinement head's U-Net
    ('three_levels_input_conv', [16, 32, 48], 6, 60000, [128, 96, 48], 2),
    ('one_level', [32], None, 3000, [24, 24, 24], 3),
]


@pytest.mark.parametrize('name,planes,cin,n,shape,batch', CASES, ids=[c[0] for c in CASES])
def test_train_executor_equals_module_path(name, planes, cin, n, shape, batch):
    rng = np.random.default_rng(len(planes) + n)
    idx = _voxels(rng, n, shape, batch)
    M = idx.shape[0]
    torch.manual_seed(3)
    ref = Net(planes, cin).to(DEV).train()
    _randomise(ref, 5)
    net = copy.deepcopy(ref)
    x0 = torch.randn(M, cin if cin is not None else planes[0], device=DEV)
    g_out = torch.randn(M, planes[0], device=DEV)

    xr = x0.clone().requires_grad_(True)
    out_r = ref(spconv.SparseConvTensor(xr, idx, shape, batch))
    out_r.backward(g_out)

    ex = UNetTrainExecutor(net.unet, net.input_conv, net.output_layer)
    xe = x0.clone().requires_grad_(True)
    assert ex.usable(xe)
    out_e = ex(spconv.SparseConvTensor(xe, idx, shape, batch))
    out_e.backward(g_out)

    errs = dict(out=_close(out_e.detach(), out_r.detach(), 'output'),
                g_in=_close_l2(xe.grad, xr.grad, 'input gradient'))
    worst = ('', 0.0)
    for (k, pe), (_, pr) in zip(net.named_parameters(), ref.named_parameters()):
        assert pe.grad is not None, k
        e = _close_l2(pe.grad, pr.grad, f'gradient of {k}')
        worst = max(worst, (k, e), key=lambda kv: kv[1])
    for (k, be), (_, br) in zip(net.named_buffers(), ref.named_buffers()):
        if k.endswith('num_batches_tracked'):
            assert int(be) == int(br) == 1, k
        else:
            _close(be, br, f'buffer {k}', 1e-5)
    print(f'{name}: {M} voxels; output {errs["out"]:.1e}, input gradient {errs["g_in"]:.1e}, '
          f'worst parameter gradient {worst[1]:.1e} ({worst[0]})')


def test_train_executor_is_deterministic_and_reusable():
    """two steps on the same executor (arena reuse) give bit-identical results for identical inputs"""
    rng = np.random.default_rng(1)
    shape, batch = [20, 20, 20], 30
    idx = _voxels(rng, 7000, shape, batch)
    M = idx.shape[0]
    torch.manual_seed(0)
    net = Net([32, 64]).to(DEV).train()
    ex = UNetTrainExecutor(net.unet, None, net.output_layer)
    x = torch.randn(M, 32, device=DEV)
    g = torch.randn(M, 32, device=DEV)
    runs = []
    for _ in range(3):
        for m in net.modules():
            if isinstance(m, nn.BatchNorm1d):
                m.reset_running_stats()
        net.zero_grad(set_to_none=True)
        out = ex(spconv.SparseConvTensor(x, idx, shape, batch))
        out.backward(g)
        runs.append([out.detach().clone()] + [p.grad.clone() for p in net.parameters()])
    for r in runs[1:]:
        for a, b in zip(r, runs[0]):
            assert torch.equal(a, b)


def test_train_executor_skips_frozen_parameters_and_input():
    rng = np.random.default_rng(2)
    shape, batch = [20, 20, 20], 10
    idx = _voxels(rng, 4000, shape, batch)
    M = idx.shape[0]
    torch.manual_seed(1)
    ref = Net([32, 64]).to(DEV).train()
    frozen = [n for i, (n, _) in enumerate(ref.named_parameters()) if i % 3 == 0]
    for n, p in ref.named_parameters():
        p.requires_grad_(n not in frozen)
    net = copy.deepcopy(ref)
    x = torch.randn(M, 32, device=DEV)                  # no gradient wanted for the input
    g = torch.randn(M, 32, device=DEV)
    ref(spconv.SparseConvTensor(x, idx, shape, batch)).backward(g)
    ex = UNetTrainExecutor(net.unet, None, net.output_layer)
    assert ex.usable(x)
    ex(spconv.SparseConvTensor(x, idx, shape, batch)).backward(g)
    for (k, pe), (_, pr) in zip(net.named_parameters(), ref.named_parameters()):
        if k in frozen:
            assert pe.grad is None and pr.grad is None, k
        else:
            _close_l2(pe.grad, pr.grad, f'gradient of {k}')


def _agg_l2(pairs):
    num = sum(float((a.double() - b.double()).pow(2).sum()) for a, b in pairs)
    den = sum(float(b.double().pow(2).sum()) for _, b in pairs)
    return (num / max(den, 1e-300)) ** 0.5


def test_train_executor_under_bf16_autocast():
    """bf16 autocast: the executor rounds the conv operands to bf16 (one MFMA per product) and keeps
    fp32 sums and activations; the module path also stores bf16 activations.  Both are bf16-accurate
    versions of the fp32 step: the executor must be at least as close to the fp32 module path as the
    module path under autocast is (all gradients taken together, relative L2)."""
    rng = np.random.default_rng(4)
    shape, batch = [20, 20, 20], 40
    idx = _voxels(rng, 9000, shape, batch)
    M = idx.shape[0]
    torch.manual_seed(2)
    ref = Net([32, 64]).to(DEV).train()
    net, mod = copy.deepcopy(ref), copy.deepcopy(ref)
    x = torch.randn(M, 32, device=DEV)
    g = torch.randn(M, 32, device=DEV)
    ref(spconv.SparseConvTensor(x, idx, shape, batch)).backward(g)
    ex = UNetTrainExecutor(net.unet, None, net.output_layer)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out = ex(spconv.SparseConvTensor(x, idx, shape, batch))
        out_m = mod(spconv.SparseConvTensor(x, idx, shape, batch))
    assert out.dtype == torch.float32
    out.backward(g)
    out_m.float().backward(g)
    e_exec = _agg_l2([(pe.grad, pr.grad) for pe, pr in zip(net.parameters(), ref.parameters())])
    e_mod = _agg_l2([(pm.grad, pr.grad) for pm, pr in zip(mod.parameters(), ref.parameters())])
    print(f'bf16 autocast, all gradients, relative L2 to the fp32 step: executor {e_exec:.3e}, modules {e_mod:.3e}')
    assert 1e-4 < e_exec <= max(1.25 * e_mod, 2e-2), (e_exec, e_mod)


@pytest.mark.parametrize('case', ['s3dis_fold5', 'scannet_full'])
def test_forward_train_with_and_without_the_executor(case):
    """SoftGroup.forward_train on the golden training batches (tests/golden/make_ref_train.py;
    s3dis_fold5: frozen backbone, the tiny U-Net trains; scannet_full: nothing frozen, backbone AND
    tiny U-Net on the native executor) against the same model on the modules: same losses, same
    gradients"""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_train_gpu import _train_case
    model, batch, ref, seed = _train_case(case)
    other = copy.deepcopy(model)
    model.train()
    other.train()
    other.use_train_executor = False
    res = []
    for m in (model, other):
        torch.manual_seed(seed)
        loss, log_vars = m(batch, return_loss=True)
        loss.backward()
        res.append((log_vars, {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    assert model.__dict__.get('_tiny_train_exec') is not None and other.__dict__.get('_tiny_train_exec') is None
    assert (model.__dict__.get('_backbone_train_exec') is not None) == (case == 'scannet_full')
    (lv_a, g_a), (lv_b, g_b) = res
    assert lv_a['num_pos'] == lv_b['num_pos'] and lv_a['num_neg'] == lv_b['num_neg']
    for k in lv_b:
        assert abs(lv_a[k] - lv_b[k]) <= 1e-4 * max(abs(lv_b[k]), 1e-3), (k, lv_a[k], lv_b[k])
    assert set(g_a) == set(g_b)
    assert any(k.startswith('unet.' if case == 'scannet_full' else 'tiny_unet.') for k in g_a)
    agg = _agg_l2([(g_a[k], g_b[k]) for k in g_b])
    print(f'{case}: {len(g_b)} gradients, all together relative L2 {agg:.2e}')
    assert agg <= 5e-3, agg
    big = max(float(v.norm()) for v in g_b.values())
    for k in g_b:
        # (the bias of a Linear in front of a batch-statistics BatchNorm has a mathematically zero
        # gradient: rounding noise on both sides, nothing to compare)
        if float(g_b[k].norm()) >= 1e-4 * big:
            _close_l2(g_a[k], g_b[k], f'gradient of {k}', 5e-2)


def test_train_executor_lifetimes():
    """tapes and arenas: two forwards before their backwards (each keeps its own arena), a forward
    whose backward never runs (tape released with the graph, arena returned), and a second backward
    through the same forward (refused: the tape is consumed)"""
    rng = np.random.default_rng(6)
    shape, batch = [20, 20, 20], 12
    idx = _voxels(rng, 5000, shape, batch)
    M = idx.shape[0]
    torch.manual_seed(4)
    ref = Net([32, 64]).to(DEV).train()
    net = copy.deepcopy(ref)
    ex = UNetTrainExecutor(net.unet, None, net.output_layer)
    xa, xb = torch.randn(M, 32, device=DEV), torch.randn(M, 32, device=DEV)
    ga, gb = torch.randn(M, 32, device=DEV), torch.randn(M, 32, device=DEV)
    # reference: the same two forwards and backwards on the modules (BatchNorm statistics update twice)
    ra = ref(spconv.SparseConvTensor(xa, idx, shape, batch))
    rb = ref(spconv.SparseConvTensor(xb, idx, shape, batch))
    rb.backward(gb)
    ra.backward(ga)
    oa = ex(spconv.SparseConvTensor(xa, idx, shape, batch))
    ob = ex(spconv.SparseConvTensor(xb, idx, shape, batch))      # first tape still alive: a second arena
    ob.backward(gb)
    oa.backward(ga)
    assert len(ex._free_arenas) == 2
    _close(oa.detach(), ra.detach(), 'first output')
    _close(ob.detach(), rb.detach(), 'second output')
    for (k, pe), (_, pr) in zip(net.named_parameters(), ref.named_parameters()):
        _close_l2(pe.grad, pr.grad, f'accumulated gradient of {k}')
    # a forward that is never differentiated
    o = ex(spconv.SparseConvTensor(xa, idx, shape, batch))
    del o
    import gc
    gc.collect()
    assert len(ex._free_arenas) == 2
    # the tape is single-use
    o = ex(spconv.SparseConvTensor(xa, idx, shape, batch))
    o.backward(ga, retain_graph=True)
    with pytest.raises(RuntimeError, match='twice'):
        o.backward(ga)


def test_fp16_autocast_with_grad_scaler_steps():
    """the reference's `fp16: True` switch (tools/train.py:47,55-62: torch.cuda.amp.autocast +
    GradScaler): dense layers run in fp16, the sparse path stays fp32 (its kernels take fp32 or bf16),
    the scaled loss back-propagates through the native executor, the scaler unscales and steps"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_train_gpu import _train_case
    model, batch, ref, seed = _train_case('s3dis_fold5')
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    before = [p.detach().clone() for p in params]
    opt = torch.optim.Adam(params, lr=1e-3)
    scaler = torch.amp.GradScaler('cuda')
    torch.manual_seed(seed)
    with torch.autocast('cuda', dtype=torch.float16):
        loss, log_vars = model(batch, return_loss=True)
    assert model.__dict__.get('_tiny_train_exec') is not None
    opt.zero_grad()
    scaler.scale(loss).backward()
    scaler.step(opt)
    scaler.update()
    assert np.isfinite(float(loss)) and abs(log_vars['loss'] - ref['loss']) <= 2e-2 * abs(ref['loss'])
    assert all(torch.isfinite(p.grad).all() for p in params if p.grad is not None)
    assert scaler.get_scale() >= 1.0
    moved = sum(float((p.detach() - b).abs().sum()) for p, b in zip(params, before))
    assert moved > 0, 'the optimizer step was skipped (non-finite gradients)'
