"""Multi-layer conv launches ("conv chain", csrc/spconv_conv.hip: conv_chain_kernel): the layers of the deep
U-Net levels (softgroup/model/blocks.py:82-143 from the strided conv into a level of <= 6144 rows down
to the deepest level and back up) and a whole small U-Net (the tiny U-Net, softgroup/model/softgroup.py:93-95)
run as ONE persistent launch per <= 22 layers with a grid barrier between two layers.
  * chains on (sg_spconv_set_chain(1)) against chains off (the default -- measured neutral to slower,
    profiles/r06_conv_chain.txt: every layer its own launch, same decomposition): BIT-identical U-Net outputs -- backbone of the ScanNet model (7 levels, chains cut into
    several launches), a 3-level U-Net with an input conv at degenerate sizes, a U-Net that is one chain from
    its first layer on (with the stray BatchNorm+ReLU step), bf16-operand arithmetic;
  * against the module path (one Python call per layer, the decomposition of the single launches): conv tolerance;
  * the whole scan with chains on == chains off: every dense result and every instance bit-identical;
  * chains of concurrent scans (3 streams) never overlap (the launch orders them) and stay bit-identical:
    test_concurrent_scans_with_chains below."""
import ctypes as C
import functools

import numpy as np
import pytest
import torch
from torch import nn

import softgroup_amd.spconv.pytorch as spconv
from softgroup_amd import _lib as L
from softgroup_amd import ops, synthetic

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _stats():
    a, b = C.c_int64(0), C.c_int64(0)
    L.check(L.lib().sg_spconv_chain_stats(C.byref(a), C.byref(b)), 'sg_spconv_chain_stats')
    return a.value, b.value


@pytest.fixture()
def chain_modes():
    lib = L.lib()

    def run(fn):
        """fn() with chains on, then off -> (on, off, chain launches, chain steps of the 'on' call)"""
        try:
            L.check(lib.sg_spconv_set_chain(1), 'sg_spconv_set_chain')
            s0 = _stats()
            on = fn()
            torch.cuda.synchronize()
            s1 = _stats()
            L.check(lib.sg_spconv_set_chain(0), 'sg_spconv_set_chain')
            off = fn()
            torch.cuda.synchronize()
            assert _stats() == s1, 'chains switched off must not launch the chain kernel'
        finally:
            lib.sg_spconv_set_chain(-1)
        return on, off, s1[0] - s0[0], s1[1] - s0[1]
    return run


def _scene(rng, n, extent, B=1):
    pts = rng.random((n, 3)) * extent
    pts[:, 2] = (np.sin(pts[:, 0] * 0.3) + np.cos(pts[:, 1] * 0.2)) * 3 + extent[2] / 2 + rng.normal(0, 0.6, n)
    v = np.clip(np.floor(pts), 0, np.array(extent) - 1).astype(np.int64)
    b = np.sort(rng.integers(0, B, n))
    key = ((b * extent[0] + v[:, 0]) * extent[1] + v[:, 1]) * extent[2] + v[:, 2]
    _, first = np.unique(key, return_index=True)
    first = np.sort(first)
    return np.concatenate([b[first, None], v[first]], 1).astype(np.int32)


def _randomise_bn(mods):
    for m in mods:
        if isinstance(m, nn.BatchNorm1d):
            with torch.no_grad():
                m.running_mean.normal_(0, 0.3)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.normal_(1, 0.2)
                m.bias.normal_(0, 0.2)


@pytest.mark.parametrize('points', [150000, 40000])
def test_backbone_chain_is_bit_identical_to_single_launches(chain_modes, points):
    xyz, rgb, inst = synthetic.scene_s2(seed=2, n=points)
    b = synthetic.make_batch(xyz, rgb, instance_labels=inst)
    b = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    model = synthetic.build_model(seed=0)
    vf = ops.voxelization(torch.cat((b['feats'], b['coords_float']), 1), b['p2v_map'])

    def fwd():
        with torch.no_grad():
            x = spconv.SparseConvTensor(vf, b['voxel_coords'].int(), b['spatial_shape'], 1)
            return [t.clone() for t in model.forward_backbone(x, b['v2p_map'])]

    on, off, launches, steps = chain_modes(fwd)
    for a, c in zip(on, off):
        assert torch.isfinite(a).all() and torch.equal(a, c), float((a - c).abs().max())
    # levels 3..6 of the 150 k scene: 43 layers + 3 concats -> more than one launch of <= 22 steps
    assert launches >= 2 and steps >= 30, (launches, steps)
    # module path (own decomposition for these layers): conv tolerance
    model.use_executor = False
    try:
        ref = fwd()
    finally:
        model.use_executor = True
    scale = float(ref[2].abs().max())
    assert float((on[2] - ref[2]).abs().max()) <= 1e-4 * max(scale, 1.0)


@pytest.mark.parametrize('n,shape', [(7, [9, 9, 9]), (700, [33, 17, 9]), (3000, [12, 12, 12]), (20000, [67, 45, 31])])
def test_small_unet_with_input_conv(chain_modes, n, shape):
    """levels that lose all their voxels, single-tile layers, fewer units than XCDs; the chain opens below
    level 0 (the input conv's 6 -> 16 padded channels are not a layer the chain takes)"""
    from softgroup_amd.model.blocks import ResidualBlock, UBlock
    from softgroup_amd.spconv import unet_exec
    torch.manual_seed(n)
    norm_fn = functools.partial(nn.BatchNorm1d, eps=1e-4, momentum=0.1)
    unet = UBlock([32, 64, 96], norm_fn, 2, ResidualBlock, indice_key_id=1).to(DEV).eval()
    inp = spconv.SparseSequential(spconv.SubMConv3d(6, 32, 3, padding=1, bias=False, indice_key='subm1')).to(DEV).eval()
    out_layer = spconv.SparseSequential(norm_fn(32), nn.ReLU()).to(DEV).eval()
    _randomise_bn(list(unet.modules()) + list(out_layer.modules()))
    idx = _scene(np.random.default_rng(n), n, shape)
    x = spconv.SparseConvTensor(torch.randn(len(idx), 6, device=DEV), torch.from_numpy(idx).to(DEV), shape, 1)
    ex = unet_exec.UNetExecutor(unet, inp, out_layer)

    def fwd():
        with torch.no_grad():
            return ex(x).clone()

    on, off, launches, steps = chain_modes(fwd)
    assert torch.isfinite(on).all() and torch.equal(on, off), float((on - off).abs().max())
    with torch.no_grad():
        ref = out_layer(unet(inp(x))).features
    np.testing.assert_allclose(on.cpu().numpy(), ref.cpu().numpy(), atol=5e-5, rtol=1e-5)
    if n >= 700:
        assert launches >= 1


@pytest.mark.parametrize('n,planes,arith', [(2500, [32, 64], 1), (900, [32, 64, 96], 1), (2500, [32, 64], 2),
                                            (5000, [64, 128], 1)])
def test_whole_unet_as_one_chain(chain_modes, n, planes, arith):
    """no input conv and few rows (the tiny U-Net's shape): every layer, the concats and the first
    BatchNorm+ReLU are steps of one chain; arith 2 = bf16 operands (the autocast arithmetic)"""
    from softgroup_amd.model.blocks import ResidualBlock, UBlock
    from softgroup_amd.spconv import unet_exec
    torch.manual_seed(n + len(planes))
    norm_fn = functools.partial(nn.BatchNorm1d, eps=1e-4, momentum=0.1)
    unet = UBlock(planes, norm_fn, 2, ResidualBlock, indice_key_id=1).to(DEV).eval()
    out_layer = spconv.SparseSequential(norm_fn(planes[0]), nn.ReLU()).to(DEV).eval()
    _randomise_bn(list(unet.modules()) + list(out_layer.modules()))
    shape = [40, 33, 21]
    idx = _scene(np.random.default_rng(n), n, shape, B=3)
    x = spconv.SparseConvTensor(torch.randn(len(idx), planes[0], device=DEV), torch.from_numpy(idx).to(DEV), shape, 3)
    ex = unet_exec.UNetExecutor(unet, None, out_layer)
    lib = L.lib()

    def fwd():
        with torch.no_grad():
            return ex(x).clone()

    try:
        if arith != 1:
            L.check(lib.sg_spconv_set_arithmetic(arith), 'sg_spconv_set_arithmetic')
        on, off, launches, steps = chain_modes(fwd)
    finally:
        lib.sg_spconv_set_arithmetic(-1)
    assert torch.isfinite(on).all() and torch.equal(on, off), float((on - off).abs().max())
    n_layers = sum(1 for m in unet.modules() if isinstance(m, (spconv.SubMConv3d, spconv.SparseConv3d,
                                                               spconv.SparseInverseConv3d)))
    # every conv, (len(planes) - 1) concats and the leading BatchNorm+ReLU
    assert steps == n_layers + len(planes) - 1 + 1, (steps, n_layers)
    assert launches == (steps + 21) // 22
    with torch.no_grad():
        ref = out_layer(unet(x)).features
    tol = 5e-5 if arith == 1 else 3e-2
    np.testing.assert_allclose(on.cpu().numpy(), ref.cpu().numpy(), atol=tol * max(1.0, float(ref.abs().max())), rtol=1e-5)


def test_whole_scan_with_chains_equals_without(chain_modes):
    xyz, rgb, inst = synthetic.scene_s2(seed=4, n=60000, room_scale=0.6)
    b = synthetic.make_batch(xyz, rgb, instance_labels=inst)
    b = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    model = synthetic.build_model(seed=0)
    model.async_results = False

    def fwd():
        with torch.no_grad():
            return dict(model(b))

    on, off, launches, steps = chain_modes(fwd)
    assert launches >= 3          # backbone tail (>= 2 launches) + tiny U-Net
    assert len(on['pred_instances']) == len(off['pred_instances']) > 0
    for a, c in zip(on['pred_instances'], off['pred_instances']):
        assert a['label_id'] == c['label_id'] and a['conf'] == c['conf'] and a['pred_mask'] == c['pred_mask']
    for k in ('semantic_preds', 'offset_preds'):
        np.testing.assert_array_equal(on[k], off[k])


def test_concurrent_scans_with_chains():
    """three scans in flight on three streams, chains on: chain launches of different streams are ordered by
    the library (their workgroups must all be resident at once), results equal the one-at-a-time results"""
    lib = L.lib()
    scenes = []
    for i in range(3):
        xyz, rgb, inst = synthetic.scene_s2(seed=11 + i, n=50000, room_scale=0.55)
        b = synthetic.make_batch(xyz, rgb, instance_labels=inst)
        scenes.append({k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()})
    model = synthetic.build_model(seed=0)
    try:
        L.check(lib.sg_spconv_set_chain(1), 'sg_spconv_set_chain')
        with torch.no_grad():
            model.scan_contexts = 1
            alone = [dict(model(b)) for b in scenes]
            model.scan_contexts = 3
            s0 = _stats()
            rs = [model(scenes[i % 3]) for i in range(12)]
            got = [dict(r) for r in rs]
            assert _stats()[0] - s0[0] >= 12 * 3
    finally:
        lib.sg_spconv_set_chain(-1)
        model.scan_contexts = 1
    for i, g in enumerate(got):
        a = alone[i % 3]
        assert len(g['pred_instances']) == len(a['pred_instances'])
        for x, y in zip(g['pred_instances'], a['pred_instances']):
            assert x['label_id'] == y['label_id'] and x['conf'] == y['conf'] and x['pred_mask'] == y['pred_mask']
        np.testing.assert_array_equal(g['semantic_preds'], a['semantic_preds'])
        np.testing.assert_array_equal(g['offset_preds'], a['offset_preds'])
