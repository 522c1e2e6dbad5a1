"""GPU parity tests of the sparse-conv layer (spconv.pytorch subset): rulebooks bit-exact vs the
CPU oracle, conv outputs within 1e-4 (fp32; the summation order differs by design) vs the oracle,
the dense torch.nn.functional golden vectors and an unfused PyTorch fp32 reference."""
import numpy as np
import pytest
import torch
from torch import nn

import oracle
import softgroup_amd.spconv.pytorch as spconv
from softgroup_amd.spconv import core

pytestmark = pytest.mark.gpu
DEV = 'cuda'
TOL = dict(atol=1e-4, rtol=1e-4)   # north-star tolerance for float features


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _scene(rng, n, extent, B=1):
    """random surface-ish active set: unique voxels in first-seen order"""
    pts = rng.random((n, 3)) * extent
    pts[:, 2] = (np.sin(pts[:, 0] * 0.3) + np.cos(pts[:, 1] * 0.2)) * 3 + extent[2] / 2 + rng.normal(0, 0.6, n)
    v = np.clip(np.floor(pts), 0, np.array(extent) - 1).astype(np.int64)
    b = np.sort(rng.integers(0, B, n))
    key = ((b * extent[0] + v[:, 0]) * extent[1] + v[:, 1]) * extent[2] + v[:, 2]
    _, first = np.unique(key, return_index=True)
    first = np.sort(first)
    return np.concatenate([b[first, None], v[first]], 1).astype(np.int32)


def test_subm_rulebook_and_plan_exact():
    rng = np.random.default_rng(0)
    shape = [131, 97, 41]
    idx = _scene(rng, 60000, shape, B=2)
    rule = core.SubMRule(t(idx), shape)
    nbr = rule.plan.nbr.cpu().numpy()
    assert np.array_equal(nbr, oracle.subm_rulebook(idx, shape))
    # tile plan: T tiles of 32 rows (rows sorted by neighbour mask), tiles emitted heaviest first
    M = len(idx)
    T = (M + 31) // 32
    order = rule.plan.order.cpu().numpy().reshape(T, 32)
    valid = order >= 0
    assert valid.sum() == M and np.array_equal(np.sort(order[valid]), np.arange(M))
    mask = ((nbr >= 0) << np.arange(27)).sum(1).astype(np.uint32)
    row_mask = np.where(valid, mask[np.clip(order, 0, None)], 0).astype(np.uint32)
    tm = rule.plan.tile_mask.cpu().numpy().view(np.uint32)[:T]
    assert np.array_equal(tm, np.bitwise_or.reduce(row_mask, 1))
    pop = np.array([bin(int(x)).count('1') for x in tm])
    assert (np.diff(pop) <= 0).all()                                   # heaviest tiles first
    hist = rule.plan.tile_mask.cpu().numpy().view(np.uint32)[T:]
    assert np.array_equal(hist[:33], np.bincount(pop, minlength=33)) and (hist[33:] == 0).all()
    # rows are sorted by the mask with its bits permuted by offset frequency (rarest offset = MSB,
    # ties: lower offset more common); every tile is a contiguous run of that sequence, so the
    # per-tile key ranges are disjoint and sorted
    freq = ((nbr >= 0).sum(0)).astype(np.int64)
    pos = np.array([sum(1 for o in range(27) if freq[o] > freq[k] or (freq[o] == freq[k] and o < k))
                    for k in range(27)])
    key = (((mask[:, None] >> np.arange(27)) & 1).astype(np.uint64) << pos.astype(np.uint64)).sum(1)
    row_key = np.where(valid, key[np.clip(order, 0, None)], 0).astype(np.uint64)
    lo = np.where(valid, row_key, np.uint64(0xffffffff)).min(1)
    hi = row_key.max(1)
    o = np.argsort(lo, kind='stable')
    assert (hi[o][:-1] <= lo[o][1:]).all()
    # and it pays: fewer (tile, offset) pairs than sorting by the raw mask
    def pairs(seq):
        pad = np.zeros(T * 32, np.uint32)
        pad[:M] = seq
        return sum(bin(int(x)).count('1') for x in np.bitwise_or.reduce(pad.reshape(T, 32), 1))
    assert pop.sum() <= pairs(np.sort(mask))
    nt = rule.plan.nbr_tiles.cpu().numpy().reshape(T, 32, 27)
    exp = np.where(valid[:, :, None], nbr[np.clip(order, 0, None)], -1)
    assert np.array_equal(nt, exp)


def test_down_rulebook_exact_incl_odd_extent_drop():
    rng = np.random.default_rng(1)
    shape = [129, 65, 33]                           # odd: last plane of every axis is dropped
    idx = _scene(rng, 40000, shape, B=3)
    idx[:50, 1] = 128
    rule = core.DownRule(t(idx), shape, 3)
    oi, in2out, child, oshape = oracle.down_rulebook(idx, shape)
    assert rule.out_spatial_shape == oshape == [64, 32, 16]
    assert np.array_equal(rule.in2out.cpu().numpy(), in2out)
    assert np.array_equal(rule.out_indices.cpu().numpy(), oi)
    assert np.array_equal(rule.plan.nbr.cpu().numpy(), child)
    assert (in2out[:50] == -1).all()
    inv = rule.inv_plan.nbr.cpu().numpy()
    k = (idx[:, 1] & 1) * 4 + (idx[:, 2] & 1) * 2 + (idx[:, 3] & 1)
    exp = np.full((len(idx), 8), -1, np.int32)
    exp[np.arange(len(idx)), k] = in2out
    assert np.array_equal(inv, exp)


HIST = 40      # SG_PLAN_HIST_WORDS: the plan's tile-weight histogram behind its T tile masks


def _pyramid(idx, shape, n_levels):
    """sg_spconv_pyramid_rows + _build through the C ABI -> per-level dict of numpy arrays"""
    import ctypes as C
    from softgroup_amd import _lib as L
    lib = L.lib()

    class PlanPtrs(C.Structure):
        _fields_ = [('order', C.c_void_p), ('tile_mask', C.c_void_p), ('nbr_tiles', C.c_void_p)]

    class Level(C.Structure):
        _fields_ = [('rows', C.c_int), ('indices', C.c_void_p), ('nbr', C.c_void_p), ('subm', PlanPtrs),
                    ('in2out', C.c_void_p), ('child', C.c_void_p), ('down', PlanPtrs),
                    ('inv', C.c_void_p), ('up', PlanPtrs)]

    M0 = len(idx)
    d_idx = t(idx)
    shp = (C.c_int32 * 3)(*shape)
    ws = L.workspace(lib.sg_spconv_pyramid_workspace_bytes(M0, n_levels), DEV)
    rows_dev = torch.full((n_levels, ), -7, dtype=torch.int32, device=DEV)
    L.check(lib.sg_spconv_pyramid_rows(L.ptr(d_idx), M0, shp, n_levels, L.ptr(rows_dev), L.ptr(ws),
                                       ws.numel(), L.stream()), 'sg_spconv_pyramid_rows')
    rows = rows_dev.tolist()
    lv = (Level * n_levels)()
    keep = []

    def buf(n, dtype=torch.int32):
        b = torch.full((max(int(n), 1), ), -99, dtype=dtype, device=DEV)
        keep.append(b)
        return b

    out = []
    for l in range(n_levels):
        r, r2 = rows[l], rows[l + 1] if l + 1 < n_levels else 0
        T, T2 = (r + 31) // 32, (r2 + 31) // 32
        d = dict(rows=r, indices=buf(r * 4), nbr=buf(r * 27), subm=(buf(T * 32), buf(T + HIST), buf(T * 32 * 27)))
        lv[l].rows = r
        lv[l].indices, lv[l].nbr = d['indices'].data_ptr(), d['nbr'].data_ptr()
        lv[l].subm = PlanPtrs(*[x.data_ptr() for x in d['subm']])
        if l + 1 < n_levels:
            d.update(in2out=buf(r), child=buf(r2 * 8), inv=buf(r * 8),
                     down=(buf(T2 * 32), buf(T2 + HIST), buf(T2 * 32 * 8)),
                     up=(buf(T * 32), buf(T + HIST), buf(T * 32 * 8)))
            lv[l].in2out, lv[l].child, lv[l].inv = (d[k].data_ptr() for k in ('in2out', 'child', 'inv'))
            lv[l].down = PlanPtrs(*[x.data_ptr() for x in d['down']])
            lv[l].up = PlanPtrs(*[x.data_ptr() for x in d['up']])
        out.append(d)
    ws2 = L.workspace(lib.sg_spconv_pyramid_build_workspace_bytes(C.byref(lv), n_levels), DEV)
    L.check(lib.sg_spconv_pyramid_build(L.ptr(d_idx), M0, shp, n_levels, C.byref(lv), L.ptr(ws), ws.numel(),
                                        L.ptr(ws2), ws2.numel(), L.stream()), 'sg_spconv_pyramid_build')
    torch.cuda.synchronize()
    return rows, out


def _check_plan(plan, nbr, rows, K):
    """a tile plan is valid for its table and equals the per-level planner's (same keys, stable)"""
    order, tmask, ntiles = (x.cpu().numpy() for x in plan)
    T = (rows + 31) // 32
    if rows == 0:
        return
    order = order[:T * 32].reshape(T, 32)
    valid = order >= 0
    assert valid.sum() == rows and np.array_equal(np.sort(order[valid]), np.arange(rows))
    mask = ((nbr >= 0) << np.arange(K)).sum(1).astype(np.uint32)
    row_mask = np.where(valid, mask[np.clip(order, 0, None)], 0).astype(np.uint32)
    tm = tmask[:T].view(np.uint32)
    assert np.array_equal(tm, np.bitwise_or.reduce(row_mask, 1))
    pop = np.array([bin(int(x)).count('1') for x in tm])
    assert (np.diff(pop) <= 0).all()                                    # heaviest tiles first
    # histogram of the tile weights behind the masks (hist[j] = tiles with j offsets)
    hist = tmask[T:T + HIST].view(np.uint32)
    assert np.array_equal(hist[:33], np.bincount(pop, minlength=33)) and (hist[33:] == 0).all()
    exp = np.where(valid[:, :, None], nbr[np.clip(order, 0, None)], -1)
    assert np.array_equal(ntiles[:T * 32 * K].reshape(T, 32, K), exp)
    # identical row sequence to sg_spconv_plan (tiles of equal weight may be emitted in another order)
    ref = core._Plan(t(nbr.astype(np.int32)), rows, K)
    ro = ref.order.cpu().numpy().reshape(T, 32)
    assert sorted(map(tuple, ro.tolist())) == sorted(map(tuple, order.tolist()))


def test_whole_pyramid_index_build_equals_the_per_level_chain():
    """sg_spconv_pyramid_rows / _build (all levels of a U-Net in a handful of launches, what
    sg_unet_forward uses) against the per-level chain of the oracle: row counts, coordinates,
    in2out / child / inverse tables, SubM tables bit-exact -- including the odd-extent drops at
    every level -- and every tile plan valid and equal to the per-level planner's."""
    rng = np.random.default_rng(4)
    for shape, n, B, n_levels in (([129, 67, 35], 40000, 3, 5), ([300, 250, 135], 90000, 1, 7),
                                  ([20, 20, 20], 3000, 7, 2), ([5, 3, 2], 30, 1, 4)):
        idx = _scene(rng, n, shape, B=B)
        rows, lv = _pyramid(idx, shape, n_levels)
        cur, sh = idx, list(shape)
        for l in range(n_levels):
            d = lv[l]
            assert rows[l] == len(cur), (shape, l, rows, len(cur))
            if len(cur) == 0:
                break
            assert np.array_equal(d['indices'].cpu().numpy()[:len(cur) * 4].reshape(-1, 4), cur)
            nbr = oracle.subm_rulebook(cur, sh)
            assert np.array_equal(d['nbr'].cpu().numpy()[:len(cur) * 27].reshape(-1, 27), nbr)
            _check_plan(d['subm'], nbr, len(cur), 27)
            if l + 1 == n_levels:
                break
            oi, in2out, child, osh = oracle.down_rulebook(cur, sh)
            assert np.array_equal(d['in2out'].cpu().numpy()[:len(cur)], in2out)
            assert np.array_equal(d['child'].cpu().numpy()[:len(oi) * 8].reshape(-1, 8), child)
            k = (cur[:, 1] & 1) * 4 + (cur[:, 2] & 1) * 2 + (cur[:, 3] & 1)
            inv = np.full((len(cur), 8), -1, np.int32)
            inv[np.arange(len(cur)), k] = in2out
            assert np.array_equal(d['inv'].cpu().numpy()[:len(cur) * 8].reshape(-1, 8), inv)
            _check_plan(d['down'], child, len(oi), 8)
            _check_plan(d['up'], inv, len(cur), 8)
            cur, sh = oi, osh


@pytest.mark.parametrize('cin,cout', [(6, 32), (32, 32), (64, 32), (64, 64), (96, 224), (32, 16), (3, 48)])
def test_subm_conv_vs_oracle(cin, cout):
    rng = np.random.default_rng(cin * 1000 + cout)
    shape = [64, 64, 32]
    idx = _scene(rng, 9000, shape)
    f = rng.standard_normal((len(idx), cin)).astype(np.float32)
    conv = spconv.SubMConv3d(cin, cout, kernel_size=3, padding=1, bias=False, indice_key='s').to(DEV)
    with torch.no_grad():
        out = conv(spconv.SparseConvTensor(t(f), t(idx), shape, 1))
    ref = oracle.subm_conv3d(f, oracle.subm_rulebook(idx, shape), conv.weight.detach().cpu().numpy())
    np.testing.assert_allclose(out.features.cpu().numpy(), ref, **TOL)
    assert out.indices.data_ptr() == out.indices.data_ptr() and out.spatial_shape == shape


def test_conv_golden_dense_equivalence(golden):
    """SubM / strided / inverse against torch.nn.functional conv3d / conv_transpose3d on the dense
    grid (fixtures from tests/golden/make_golden.py); extent 9 is odd -> last plane dropped."""
    g = golden('sparse_conv_dense')
    idx, f, shape = g['indices'], g['feats'], [int(s) for s in g['shape']]
    x = spconv.SparseConvTensor(t(f), t(idx), shape, 2)
    subm = spconv.SubMConv3d(32, 64, 3, padding=1, bias=False, indice_key='subm1').to(DEV)
    down = spconv.SparseConv3d(32, 64, kernel_size=2, stride=2, bias=False, indice_key='spconv1').to(DEV)
    inv = spconv.SparseInverseConv3d(64, 32, kernel_size=2, bias=False, indice_key='spconv1').to(DEV)
    with torch.no_grad():
        subm.weight.copy_(t(g['W_subm']))
        down.weight.copy_(t(g['W_down']))
        inv.weight.copy_(t(g['W_inv']))
        y = subm(x)
        np.testing.assert_allclose(y.features.cpu().numpy(), g['subm_out'], **TOL)
        d = down(x)
        assert d.spatial_shape == [4, 4, 4]
        oi = d.indices.cpu().numpy()
        np.testing.assert_allclose(d.features.cpu().numpy(),
                                   g['down_dense'][oi[:, 0], oi[:, 1], oi[:, 2], oi[:, 3]], **TOL)
        u = inv(d)
        assert u.indices.data_ptr() == x.indices.data_ptr() and u.spatial_shape == shape
        np.testing.assert_allclose(u.features.cpu().numpy(), g['inverse_out'], **TOL)


def test_fused_bn_relu_residual_matches_unfused_torch():
    rng = np.random.default_rng(9)
    shape = [48, 48, 48]
    idx = _scene(rng, 7000, shape)
    f = rng.standard_normal((len(idx), 64)).astype(np.float32)
    bn = nn.BatchNorm1d(64, eps=1e-4).to(DEV)
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.3)
        bn.running_var.uniform_(0.5, 1.5)
        bn.weight.normal_(1, 0.2)
        bn.bias.normal_(0, 0.2)
    conv = spconv.SubMConv3d(64, 32, 3, padding=1, bias=False, indice_key='k').to(DEV)
    seq = spconv.SparseSequential(bn, nn.ReLU(), conv).eval()
    x = spconv.SparseConvTensor(t(f), t(idx), shape, 1)
    res = torch.randn(len(idx), 32, device=DEV)
    with torch.no_grad():
        fused = seq(x, residual=res).features
        act = torch.relu(bn(x.features))
        unf = conv(x.replace_feature(act)).features + res
    np.testing.assert_allclose(fused.cpu().numpy(), unf.cpu().numpy(), **TOL)
    # and against the CPU oracle on the same activations
    ref = oracle.subm_conv3d(act.cpu().numpy(), oracle.subm_rulebook(idx, shape),
                             conv.weight.detach().cpu().numpy()) + res.cpu().numpy()
    np.testing.assert_allclose(fused.cpu().numpy(), ref, **TOL)
    # absent neighbours must contribute 0, not act(0): a lone voxel sees only its centre tap
    lone = spconv.SparseConvTensor(torch.zeros(1, 64, device=DEV), torch.tensor([[0, 5, 5, 5]], dtype=torch.int32, device=DEV), shape, 1)
    with torch.no_grad():
        y = seq(lone).features
        exp = torch.relu(bn(lone.features)) @ conv.weight[:, 1, 1, 1, :].T
    np.testing.assert_allclose(y.cpu().numpy(), exp.cpu().numpy(), **TOL)
    # BN -> ReLU tail (output_layer) as one elementwise kernel
    tail = spconv.SparseSequential(bn, nn.ReLU()).eval()
    with torch.no_grad():
        np.testing.assert_allclose(tail(x).features.cpu().numpy(), act.cpu().numpy(), atol=1e-6, rtol=1e-6)


def test_rulebook_shared_by_indice_key_and_empty_input():
    shape = [32, 32, 32]
    rng = np.random.default_rng(4)
    idx = _scene(rng, 2000, shape)
    x = spconv.SparseConvTensor(torch.randn(len(idx), 32, device=DEV), t(idx), shape, 1)
    a = spconv.SubMConv3d(32, 32, 3, padding=1, bias=False, indice_key='subm1').to(DEV)
    b = spconv.SubMConv3d(32, 64, 3, padding=1, bias=False, indice_key='subm1').to(DEV)
    with torch.no_grad():
        y = a(x)
        rule = x.indice_dict['subm1']
        z = b(y)
        assert y.indice_dict is x.indice_dict and z.indice_dict['subm1'] is rule
        e = spconv.SparseConvTensor(torch.zeros(0, 32, device=DEV), torch.zeros((0, 4), dtype=torch.int32, device=DEV), shape, 1)
        assert a(e).features.shape == (0, 32)
    inv = spconv.SparseInverseConv3d(32, 32, 2, bias=False, indice_key='nokey').to(DEV)
    with pytest.raises(RuntimeError):
        with torch.no_grad():
            inv(x)
    with pytest.raises(NotImplementedError):
        spconv.SparseConv3d(8, 8, kernel_size=3, stride=2)


def test_large_scene_linearity_and_determinism():
    """full-size property checks (sizes the oracle would take long on): conv(a*x + b*y) =
    a*conv(x) + b*conv(y), and two runs are bit-identical (output-stationary, no atomics)."""
    rng = np.random.default_rng(12)
    shape = [320, 270, 150]
    idx = _scene(rng, 200000, shape)
    M = len(idx)
    conv = spconv.SubMConv3d(64, 64, 3, padding=1, bias=False, indice_key='k').to(DEV)
    x = torch.randn(M, 64, device=DEV)
    y = torch.randn(M, 64, device=DEV)
    ti = t(idx)
    with torch.no_grad():
        cx = conv(spconv.SparseConvTensor(x, ti, shape, 1)).features
        cy = conv(spconv.SparseConvTensor(y, ti, shape, 1)).features
        cxy = conv(spconv.SparseConvTensor(2 * x - 3 * y, ti, shape, 1)).features
        cx2 = conv(spconv.SparseConvTensor(x, ti, shape, 1)).features
    assert torch.equal(cx, cx2)
    np.testing.assert_allclose(cxy.cpu().numpy(), (2 * cx - 3 * cy).cpu().numpy(), atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize('cin,cout,n', [(64, 64, 200000), (96, 96, 60000), (128, 128, 9000), (192, 192, 400),
                                        (32, 32, 200000)])
def test_split_precision_products_equal_fp32_mfma(cin, cout, n):
    """The default arithmetic of the conv kernel puts the fp32 products on the bf16 matrix pipe
    (operands split three ways, six bf16 MFMAs per slice, fp32 accumulation).  Against the fp32-MFMA
    kernel on the same layer, inputs of mixed magnitude (1e-3 .. 1e3): the difference stays at the
    level of two fp32 summation orders (<= 1e-5 of the row scale asserted, ~1e-6 measured), well
    inside the 1e-4 bar, on every launch shape (64-column units, 8-wave units, offset-split tiny
    layers, the gather-bound 32-channel layer)."""
    from softgroup_amd import _lib as L
    rng = np.random.default_rng(cin + n)
    shape = [320, 270, 150] if n > 20000 else [64, 64, 32]
    idx = _scene(rng, n, shape)
    M = len(idx)
    conv = spconv.SubMConv3d(cin, cout, 3, padding=1, bias=False, indice_key='k').to(DEV)
    x = (torch.randn(M, cin, device=DEV) * torch.exp(torch.empty(M, 1, device=DEV).uniform_(-7, 7))).contiguous()
    lib = L.lib()
    try:
        outs = []
        for mode in (0, 1):
            L.check(lib.sg_spconv_set_arithmetic(mode), 'sg_spconv_set_arithmetic')
            with torch.no_grad():
                outs.append(conv(spconv.SparseConvTensor(x, t(idx), shape, 1)).features.double())
    finally:
        lib.sg_spconv_set_arithmetic(-1)
    scale = outs[0].abs().max(1, keepdim=True)[0].clamp(min=1e-30)
    rel = ((outs[0] - outs[1]).abs() / scale).max().item()
    print(f'{cin}->{cout} x {M} rows: max |split - fp32| / row scale = {rel:.2e}')
    assert rel <= 1e-5, rel


@pytest.mark.parametrize('cin,cout,n', [(64, 64, 200000), (96, 96, 60000), (128, 128, 9000), (192, 192, 400),
                                        (32, 64, 200000), (16, 32, 120000)])
def test_bf16_operand_arithmetic_equals_fp32_kernel_on_rounded_operands(cin, cout, n):
    """sg_spconv_set_arithmetic(2) / sg_unet_desc.arithmetic = 2 (what a frozen backbone runs in under
    bf16 autocast): activations and weights rounded to nearest-even bf16, ONE MFMA per product, fp32
    sums.  The products of two bf16 numbers are exact in fp32, so the result must equal the fp32-MFMA
    kernel run on operands that were rounded beforehand, up to the order of the fp32 additions."""
    from softgroup_amd import _lib as L
    rng = np.random.default_rng(cin + n + 1)
    shape = [320, 270, 150] if n > 20000 else [64, 64, 32]
    idx = _scene(rng, n, shape)
    M = len(idx)
    conv = spconv.SubMConv3d(cin, cout, 3, padding=1, bias=False, indice_key='k').to(DEV)
    conv_r = spconv.SubMConv3d(cin, cout, 3, padding=1, bias=False, indice_key='k').to(DEV)
    with torch.no_grad():
        conv_r.weight.copy_(conv.weight.bfloat16().float())
    x = (torch.randn(M, cin, device=DEV) * torch.exp(torch.empty(M, 1, device=DEV).uniform_(-4, 4))).contiguous()
    lib = L.lib()
    try:
        with torch.no_grad():
            L.check(lib.sg_spconv_set_arithmetic(2), 'sg_spconv_set_arithmetic')
            got = conv(spconv.SparseConvTensor(x, t(idx), shape, 1)).features.double()
            L.check(lib.sg_spconv_set_arithmetic(0), 'sg_spconv_set_arithmetic')
            want = conv_r(spconv.SparseConvTensor(x.bfloat16().float(), t(idx), shape, 1)).features.double()
            exact = conv(spconv.SparseConvTensor(x, t(idx), shape, 1)).features.double()
    finally:
        lib.sg_spconv_set_arithmetic(-1)
    scale = want.abs().max(1, keepdim=True)[0].clamp(min=1e-30)
    rel = ((got - want).abs() / scale).max().item()
    gap = ((got - exact).abs() / scale).max().item()
    print(f'{cin}->{cout} x {M} rows: max |bf16 operands - fp32 on rounded| / row scale = {rel:.2e}; '
          f'distance to the fp32 result {gap:.2e}')
    assert rel <= 1e-5, rel
    assert 1e-5 < gap < 5e-2, gap        # it IS the reduced-precision arithmetic, and only that


@pytest.mark.parametrize('cin,cout', [(48, 48), (16, 112), (64, 96)])
def test_large_layer_with_a_partial_last_column_block(cin, cout):
    """>= 70 k output rows with Cout % 64 != 0 (the STPLS3D channels = 16 pyramid has 48 and 112):
    sizes where the persistent kernel's 64-column units are eligible by launch count.  They may
    only be used when every column block of a unit is whole; otherwise lanes of the second block
    past Cout would write into the next row.  Checked against plain torch on the gather table."""
    rng = np.random.default_rng(cin + cout)
    shape = [320, 270, 150]
    idx = _scene(rng, 200000, shape)
    M = len(idx)
    assert M >= 70000
    conv = spconv.SubMConv3d(cin, cout, 3, padding=1, bias=False, indice_key='k').to(DEV)
    x = torch.randn(M, cin, device=DEV)
    with torch.no_grad():
        st = spconv.SparseConvTensor(x, t(idx), shape, 1)
        got = conv(st).features
        nbr = core.SubMRule(t(idx), shape).plan.nbr.long()                  # [M, 27]
        w = conv.weight.detach().reshape(cout, 27, cin)
        xp = torch.cat([x, x.new_zeros(1, cin)])                           # row M = absent neighbour
        ref = torch.zeros(M, cout, device=DEV, dtype=torch.float64)
        for k in range(27):
            ref += xp[torch.where(nbr[:, k] >= 0, nbr[:, k], M)].double() @ w[:, k].t().double()
    np.testing.assert_allclose(got.cpu().numpy(), ref.float().cpu().numpy(), atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize('cin,cout,n,kernel', [
    (64, 64, 9000, 'persistent'),      # several rounds of units
    (128, 96, 300, 'persistent+split'),  # tiny layer: offsets split over units, reduce kernel
    (6, 32, 5000, 'tile'),             # Cin % 16 != 0: general kernel
    (32, 6, 2000, 'scalar'),           # Cout % 4 != 0: scalar path
])
def test_conv_epilogues_through_the_c_abi(cin, cout, n, kernel):
    """sg_spconv_gather_conv_f32 with every epilogue at once -- residual, in-place BatchNorm+ReLU
    (`post`) and the second activated output (`act`) -- on each kernel path, against plain torch on
    the oracle's gather table."""
    from softgroup_amd import _lib as L
    rng = np.random.default_rng(cin * 7 + cout)
    shape = [40, 40, 24]
    idx = _scene(rng, n, shape)
    M = len(idx)
    rule = core.SubMRule(t(idx), shape)
    plan = rule.plan
    f = torch.randn(M, cin, device=DEV)
    w = torch.randn(cout, 27, cin, device=DEV) * 0.1
    res = torch.randn(M, cout, device=DEV)
    ps, pb = torch.rand(cout, device=DEV) + 0.5, torch.randn(cout, device=DEV) * 0.2
    as_, ab = torch.rand(cout, device=DEV) + 0.5, torch.randn(cout, device=DEV) * 0.2
    w_k8 = core.pack_weight(w, cout, 27, cin, False)
    out = torch.empty(M, cout, device=DEV)
    out_act = torch.empty(M, cout, device=DEV)
    lib = L.lib()
    nb = lib.sg_spconv_conv_workspace_bytes(M, cout)
    ws = L.workspace(nb, f.device)
    L.check(lib.sg_spconv_gather_conv_f32(
        L.ptr(f), M, L.ptr(plan.nbr), M, 27, cin, cout, L.ptr(w_k8), L.ptr(ps), L.ptr(pb), L.ptr(res),
        L.ptr(as_), L.ptr(ab), L.ptr(out_act), L.ptr(plan.order), L.ptr(plan.tile_mask),
        L.ptr(plan.nbr_tiles), L.ptr(out), L.ptr(ws), nb, L.stream()), 'sg_spconv_gather_conv_f32')
    nbr = plan.nbr.long()
    gathered = torch.where((nbr >= 0)[:, :, None], f[nbr.clamp(min=0)], torch.zeros((), device=DEV))   # [M,27,cin]
    ref = torch.einsum('mkc,okc->mo', gathered.double(), w.double()) + res.double()
    ref = torch.relu(ref * ps.double() + pb.double())
    ref_act = torch.relu(ref * as_.double() + ab.double())
    np.testing.assert_allclose(out.cpu().numpy(), ref.float().cpu().numpy(), atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(out_act.cpu().numpy(), ref_act.float().cpu().numpy(), atol=2e-4, rtol=1e-4)


def test_unet_executor_small_net_and_arena_error():
    """sg_unet_forward on a 3-level UBlock (odd extents, channels 16/32/48) against the module
    path; an arena that is too small must come back as an error, not as a crash."""
    import functools
    from softgroup_amd import _lib as L
    from softgroup_amd.model.blocks import ResidualBlock, UBlock
    from softgroup_amd.spconv import unet_exec
    torch.manual_seed(3)
    norm_fn = functools.partial(nn.BatchNorm1d, eps=1e-4, momentum=0.1)
    unet = UBlock([16, 32, 48], norm_fn, 2, ResidualBlock, indice_key_id=1).to(DEV).eval()
    out_layer = spconv.SparseSequential(norm_fn(16), nn.ReLU()).to(DEV).eval()
    for m in list(unet.modules()) + list(out_layer.modules()):
        if isinstance(m, nn.BatchNorm1d):
            with torch.no_grad():
                m.running_mean.normal_(0, 0.3)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.normal_(1, 0.2)
                m.bias.normal_(0, 0.2)
    rng = np.random.default_rng(8)
    shape = [45, 37, 29]
    idx = _scene(rng, 6000, shape, B=2)
    x = spconv.SparseConvTensor(torch.randn(len(idx), 16, device=DEV), t(idx), shape, 2)
    ex = unet_exec.UNetExecutor(unet, None, out_layer)
    with torch.no_grad():
        assert ex.usable(x.features)
        got = ex(x)
        ref = out_layer(unet(x)).features
    np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), atol=2e-5, rtol=1e-5)
    # too small an arena
    orig = unet_exec._get_arena
    unet_exec._get_arena = lambda nb, dev: torch.empty(1 << 16, dtype=torch.uint8, device=dev)
    try:
        with torch.no_grad(), pytest.raises(L.SoftGroupHipError, match='arena'):
            ex(x)
    finally:
        unet_exec._get_arena = orig
    with torch.no_grad():
        assert torch.equal(ex(x), got)          # and the executor is still usable, bit-identical


@pytest.mark.parametrize('n,shape', [(1, [8, 8, 8]), (7, [9, 9, 9]), (40, [5, 6, 7]), (700, [33, 17, 9]),
                                     (3000, [12, 12, 12])])
def test_unet_executor_degenerate_sizes(n, shape):
    """tiny / odd inputs: levels that lose all voxels to the odd-extent drop, single-tile layers,
    fewer units than XCDs -- executor and module path must agree and stay finite."""
    import functools
    from softgroup_amd.model.blocks import ResidualBlock, UBlock
    from softgroup_amd.spconv import unet_exec
    torch.manual_seed(n)
    norm_fn = functools.partial(nn.BatchNorm1d, eps=1e-4, momentum=0.1)
    unet = UBlock([32, 64, 96], norm_fn, 2, ResidualBlock, indice_key_id=1).to(DEV).eval()
    inp = spconv.SparseSequential(spconv.SubMConv3d(6, 32, 3, padding=1, bias=False, indice_key='subm1')).to(DEV).eval()
    out_layer = spconv.SparseSequential(norm_fn(32), nn.ReLU()).to(DEV).eval()
    rng = np.random.default_rng(n)
    idx = _scene(rng, n, shape)
    x = spconv.SparseConvTensor(torch.randn(len(idx), 6, device=DEV), t(idx), shape, 1)
    ex = unet_exec.UNetExecutor(unet, inp, out_layer)
    with torch.no_grad():
        got = ex(x)
        ref = out_layer(unet(inp(x))).features
    assert torch.isfinite(got).all()
    np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), atol=5e-5, rtol=1e-5)


@pytest.mark.parametrize('cin,cout,n,split', [(128, 96, 300, 1), (192, 192, 141, 1), (224, 224, 18, 1),
                                              (160, 160, 755, 1), (64, 64, 1500, 1), (64, 64, 1500, 0)])
def test_in_launch_combine_equals_the_reduce_kernel(cin, cout, n, split):
    """Layers with few output rows split their kernel offsets over several workgroups; the partial
    sums are added by conv_reduce_kernel (default) or inside the launch by the last workgroup to
    arrive at each tile (sg_spconv_set_combine(1): write-through partial stores, one agent-scope
    counter per tile, sc1 loads).  Same fixed order, so the outputs must be IDENTICAL -- with every
    epilogue at once and without any, over repeated launches on the same stream (the counters are
    reused; a stale one would drop or double a tile), for both arithmetics."""
    from softgroup_amd import _lib as L
    rng = np.random.default_rng(cin + cout + n)
    shape = [24, 24, 16]
    idx = _scene(rng, n, shape)
    M = len(idx)
    plan = core.SubMRule(t(idx), shape).plan
    w = torch.randn(cout, 27, cin, device=DEV) * 0.1
    w_k8 = core.pack_weight(w, cout, 27, cin, False)
    res = torch.randn(M, cout, device=DEV)
    ps, pb = torch.rand(cout, device=DEV) + 0.5, torch.randn(cout, device=DEV) * 0.2
    as_, ab = torch.rand(cout, device=DEV) + 0.5, torch.randn(cout, device=DEV) * 0.2
    lib = L.lib()
    nb = lib.sg_spconv_conv_workspace_bytes(M, cout)
    assert nb > 256, 'shape does not take the offset-split path'
    ws = L.workspace(nb, DEV)
    null = None

    def run(f, epilogue):
        out = torch.full((M, cout), float('nan'), device=DEV)
        out_act = torch.full((M, cout), float('nan'), device=DEV)
        args = (L.ptr(ps), L.ptr(pb), L.ptr(res), L.ptr(as_), L.ptr(ab), L.ptr(out_act)) if epilogue \
            else (null, null, null, null, null, null)
        L.check(lib.sg_spconv_gather_conv_f32(
            L.ptr(f), M, L.ptr(plan.nbr), M, 27, cin, cout, L.ptr(w_k8), *args, L.ptr(plan.order),
            L.ptr(plan.tile_mask), L.ptr(plan.nbr_tiles), L.ptr(out), L.ptr(ws), nb, L.stream()),
            'sg_spconv_gather_conv_f32')
        return out, out_act

    feats = [torch.randn(M, cin, device=DEV) for _ in range(3)]
    try:
        L.check(lib.sg_spconv_set_arithmetic(split), 'sg_spconv_set_arithmetic')
        outs = {}
        for mode in (0, 1):
            L.check(lib.sg_spconv_set_combine(mode), 'sg_spconv_set_combine')
            outs[mode] = [run(f, ep) for f in feats for ep in (True, False)] * 1
            outs[mode] += [run(feats[0], True)]          # and once more on the reused counters
    finally:
        lib.sg_spconv_set_combine(-1)
        lib.sg_spconv_set_arithmetic(-1)
    for (a, a_act), (b, b_act) in zip(outs[0], outs[1]):
        assert torch.isfinite(b).all()
        assert torch.equal(a, b), (a - b).abs().max().item()
        if torch.isfinite(a_act).all():
            assert torch.equal(a_act, b_act)
        else:
            assert torch.isnan(b_act).all()              # no second output asked: left untouched


@pytest.mark.parametrize('points', [30000, 150000])
def test_executor_bf16_rows_under_autocast(points):
    """sg_unet_desc.arithmetic = 3 (a frozen backbone under bf16 autocast, reference tools/train.py:47): bf16
    operands AND bf16 activation rows between the layers -- 64-byte half-line gathers whose LDS-transposed chunks are
    MFMA operands as they are, results rounded to nearest even when stored.  Against arithmetic 2 (bf16 operands,
    fp32 rows) and against fp32: the extra rounding of ~80 stored activations stays at bf16 accuracy; input and
    output of the U-Net stay fp32; deterministic."""
    from softgroup_amd import synthetic, ops
    from softgroup_amd.spconv import unet_exec
    xyz, rgb, inst = synthetic.scene_s2(seed=6, n=points, room_scale=0.5 if points < 100000 else 1.0)
    b = synthetic.make_batch(xyz, rgb, instance_labels=inst)
    b = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    model = synthetic.build_model(seed=0)
    with torch.no_grad():
        vf = ops.voxelization(torch.cat((b['feats'], b['coords_float']), 1), b['p2v_map'])
        x = spconv.SparseConvTensor(vf, b['voxel_coords'].int(), b['spatial_shape'], 1)
        f32 = model._unet_features(x).clone()
        keep = unet_exec.ROWS16
        try:
            with torch.autocast('cuda', dtype=torch.bfloat16):
                unet_exec.ROWS16 = False
                a2 = model._unet_features(x).float().clone()
                unet_exec.ROWS16 = True
                a3 = model._unet_features(x).float().clone()
                a3b = model._unet_features(x).float().clone()
        finally:
            unet_exec.ROWS16 = keep
    assert torch.isfinite(a3).all() and torch.equal(a3, a3b)
    rel = lambda u, v: float((u - v).norm() / v.norm())       # noqa: E731
    e2, e3, e32 = rel(a2, f32), rel(a3, f32), rel(a3, a2)
    print(f'{points} points: relative L2 to fp32: bf16 operands {e2:.3e}, + bf16 rows {e3:.3e}; between the two {e32:.3e}')
    assert e2 <= 3e-2 and e3 <= 4e-2 and e3 <= 3 * e2 + 1e-3, (e2, e3)
