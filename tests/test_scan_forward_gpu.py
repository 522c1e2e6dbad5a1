"""``sg_scan_forward`` (csrc/scan_forward.hip): one scan of SoftGroup.forward_test (reference
softgroup/model/softgroup.py:299-361; test loop tools/test.py:145-150) as ONE C call, against the staged
path of rounds 4-5 (voxelization op -> executor -> fused heads -> torch softmax -> sg_scan_grouping ->
executor -> torch heads -> sg_scan_instances), which the other GPU tests pin to the oracle / reference.
  * its own small kernels against torch (softmax bit for bit; MLP / Linear rows <= 1e-5),
  * every dense result bit-identical, the same instances (label, RLE string; confidence <= 1e-5: the
    class / IoU heads are FMA chains here and a GEMM library there),
  * stage tensors read back from the arena: pooled voxel features, backbone output, scores, softmax,
    proposals bit-identical to the staged computation."""
import ctypes as C

import numpy as np
import pytest
import torch

from softgroup_amd import _lib as L
from softgroup_amd import ops, synthetic
from softgroup_amd.model import scan_forward as SF

pytestmark = pytest.mark.gpu


def _cuda(batch):
    return {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}


def test_softmax_rows_is_bit_identical_to_torch():
    lib = L.lib()
    g = torch.Generator().manual_seed(3)
    for rows, cols, std in ((150000, 20, 8.0), (150000, 20, 0.05), (37, 19, 3.0), (1000, 4, 2.0), (513, 32, 6.0),
                            (64, 13, 30.0)):
        x = (torch.randn(rows, cols, generator=g) * std).cuda()
        out = torch.empty_like(x)
        L.check(lib.sg_softmax_rows(L.ptr(x), rows, cols, L.ptr(out), L.stream()), 'sg_softmax_rows')
        ref = x.softmax(-1)
        assert torch.equal(out, ref), (rows, cols, std, float((out - ref).abs().max()))


def test_mlp_and_linear_rows_equal_the_modules():
    lib = L.lib()
    torch.manual_seed(1)
    model = synthetic.build_model(seed=0)
    c = model.channels
    x = torch.randn(5000, c, device='cuda')
    idx = torch.randint(0, 5000, (12000, ), device='cuda', dtype=torch.int32)
    keep = []
    m = SF._mlp(model.mask_linear, c, keep)
    out = torch.empty(idx.numel(), m.out, device='cuda')
    L.check(lib.sg_mlp_rows(L.ptr(x), L.ptr(idx), idx.numel(), c, C.byref(m), L.ptr(out), L.stream()), 'sg_mlp_rows')
    import copy
    with torch.no_grad():
        ref = model.mask_linear(x)[idx.long()]
        ref64 = copy.deepcopy(model.mask_linear).double()(x.double())[idx.long()]
    assert float((out.double() - ref64).abs().max()) <= 1e-5 * max(1.0, float(ref64.abs().max()))
    assert float((out - ref).abs().max()) <= 1e-4
    lin = SF._lin(model.cls_linear, c)
    o2 = torch.empty(x.shape[0], lin.out, device='cuda')
    L.check(lib.sg_linear_rows(L.ptr(x), x.shape[0], C.byref(lin), L.ptr(o2), L.stream()), 'sg_linear_rows')
    with torch.no_grad():
        r2 = copy.deepcopy(model.cls_linear).double()(x.double())
    assert float((o2.double() - r2).abs().max()) <= 1e-5 * max(1.0, float(r2.abs().max()))


def _compare(new, old, conf_tol=1e-5):
    assert set(new.keys()) == set(old.keys()), (sorted(new.keys()), sorted(old.keys()))
    for k in new.keys():
        if k == 'pred_instances':
            continue
        a, b = new[k], old[k]
        if isinstance(a, np.ndarray):
            assert a.dtype == b.dtype and a.shape == b.shape, k
            assert np.array_equal(a, b, equal_nan=a.dtype.kind == 'f'), k
        else:
            assert a == b, k
    pa, pb = new.get('pred_instances'), old.get('pred_instances')
    if pa is None:
        return 0
    assert len(pa) == len(pb), (len(pa), len(pb))
    worst = 0.0
    for i, (p, q) in enumerate(zip(pa, pb)):
        assert p['label_id'] == q['label_id'] and p['scan_id'] == q['scan_id'], i
        assert p['pred_mask'] == q['pred_mask'], f'instance {i}: RLE mask differs'
        worst = max(worst, abs(float(p['conf']) - float(q['conf'])))
    assert worst <= conf_tol, worst
    return len(pa)


@pytest.mark.parametrize('n_points', [30000, 150000])
def test_one_call_scan_equals_the_staged_path(n_points):
    model = synthetic.build_model(seed=0)
    model.async_results = False
    total = 0
    for seed in (1, 2, 5):
        xyz, rgb, inst = synthetic.scene_s2(seed=seed, n=n_points, room_scale=0.45 if n_points < 100000 else 1.0)
        b = _cuda(synthetic.make_batch(xyz, rgb, instance_labels=inst, scan_id=f's{seed}'))
        with torch.no_grad():
            model.use_scan_forward = True
            new = dict(model(b))
            assert model.__dict__['_scan_forward'].last.stage == 4, 'the one-call path must have run to the end'
            again = dict(model(b))
            model.use_scan_forward = False
            old = dict(model(b))
        total += _compare(new, old)
        _compare(again, new, conf_tol=0.0)          # the same call twice: bit-identical
    assert total > 50, 'scenes must exercise grouping + refinement'


def test_one_call_scan_without_proposals_and_semantic_only():
    model = synthetic.build_model(seed=0)
    model.async_results = False
    xyz, rgb = synthetic.scene_s1(seed=9, n=20000)
    b = _cuda(synthetic.make_batch(xyz, rgb, scan_id='empty'))
    with torch.no_grad():
        new = dict(model(b))
        assert model.__dict__['_scan_forward'].last.stage in (1, 2)
        model.use_scan_forward = False
        old = dict(model(b))
    assert _compare(new, old) == 0


def test_stage_tensors_of_the_one_call_scan():
    """the arena offsets of sg_scan_result: every stage against the staged computation"""
    import softgroup_amd.spconv.pytorch as spconv
    model = synthetic.build_model(seed=0)
    model.async_results = False
    xyz, rgb, inst = synthetic.scene_s2(seed=4, n=60000, room_scale=0.6)
    b = _cuda(synthetic.make_batch(xyz, rgb, instance_labels=inst))
    with torch.no_grad():
        dict(model(b))
        sf = model.__dict__['_scan_forward']
        res = sf.last
        arena = SF._arenas[(b['feats'].device, L.stream())]

        def view(off, dtype, *shape):
            n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
            return arena[off:off + n].view(dtype).view(*shape).clone()

        N, M, c, ns = b['feats'].shape[0], b['voxel_coords'].shape[0], model.channels, model.semantic_classes
        got = dict(vfeat=view(res.voxel_feats_in, torch.float32, M, 6), sem=view(res.semantic_scores, torch.float32, N, ns),
                   prob=view(res.semantic_prob, torch.float32, N, ns), off=view(res.pt_offsets, torch.float32, N, 3),
                   feats=view(res.output_feats, torch.float32, N, c), bb=view(res.backbone_out, torch.float32, M, c))
        g = res.grouping
        base = res.grouping_base
        pidx = view(base + g.proposals_idx, torch.int32, g.sum_npoint, 2)
        poff = view(base + g.proposals_offset, torch.int32, g.n_proposals + 1)
        # staged
        vf = ops.voxelization(torch.cat((b['feats'], b['coords_float']), 1), b['p2v_map'])
        x = spconv.SparseConvTensor(vf, b['voxel_coords'].int(), b['spatial_shape'], 1)
        sem, off, feats = model.forward_backbone(x, b['v2p_map'])
        rp, ro = model.forward_grouping(sem, off, b['batch_idxs'], b['coords_float'])
    assert torch.equal(got['vfeat'], vf)
    assert torch.equal(got['bb'], model._unet_features(x))
    assert torch.equal(got['sem'], sem) and torch.equal(got['off'], off) and torch.equal(got['feats'], feats)
    assert torch.equal(got['prob'], sem.softmax(-1))
    assert torch.equal(pidx.cpu(), rp.cpu().int()) and torch.equal(poff.cpu(), ro.cpu().int())
    assert g.n_proposals > 3
