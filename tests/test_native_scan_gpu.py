"""The native scan driver (csrc/scan_exec.hip: sg_scan_grouping / sg_scan_instances) against the
per-operator path it replaces -- SoftGroup.forward_grouping + clusters_voxelization +
forward_instance + get_instances on the operator surface (softgroup/model/softgroup.py:411-480,
537-604, 655-709), which the other GPU tests pin to the oracle and to the reference itself.
Integer products (proposals, voxel index, maps, labels, RLE text) must be identical, float
products bit-identical too: both paths run the same kernels on the same inputs, and the fused
glue kernels perform the same IEEE operations torch does."""
import numpy as np
import pytest
import torch

from softgroup_amd import ops, synthetic
import softgroup_amd.spconv.pytorch as spconv

pytestmark = pytest.mark.gpu


def _cuda(batch):
    return {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}


def _backbone(model, b):
    feats = torch.cat((b['feats'], b['coords_float']), 1)
    vf = ops.voxelization(feats, b['p2v_map'])
    x = spconv.SparseConvTensor(vf, b['voxel_coords'].int(), b['spatial_shape'], b['batch_size'])
    return model.forward_backbone(x, b['v2p_map'])


SCENES = [
    ('s2_30k', lambda: synthetic.scene_s2(seed=3, n=30000, room_scale=0.45)),
    ('s2_150k', lambda: synthetic.scene_s2(seed=1, n=150000)),
]


@pytest.mark.parametrize('name,make', SCENES, ids=[s[0] for s in SCENES])
def test_native_grouping_equals_operator_path(name, make):
    xyz, rgb, inst = make()
    b = _cuda(synthetic.make_batch(xyz, rgb, instance_labels=inst))
    model = synthetic.build_model(seed=0)
    with torch.no_grad():
        sem, off, out_feats = _backbone(model, b)
        # operator path
        pidx, poff = model.forward_grouping(sem, off, b['batch_idxs'], b['coords_float'], model.grouping_cfg)
        assert pidx.shape[0] > 1000 and poff.numel() > 3, 'scene must exercise the grouping head'
        it, imap = model.clusters_voxelization(pidx, poff, out_feats, b['coords_float'],
                                               **model.instance_voxel_cfg)
        _, cls_s, iou_s, mask_s = model.forward_instance(it, imap)
        model.use_native_scan = False
        preds_op = model.get_instances('s', pidx, sem, cls_s, iou_s, mask_s)
        model.use_native_scan = True
        # native driver
        assert model._native_scan_usable(sem, out_feats, False, False)
        n_pidx, n_cls, n_iou, n_mask = model._native_grouping_and_refinement(
            sem, off, b['batch_idxs'], b['coords_float'], out_feats, b['batch_size'])
        preds_nat = model.get_instances('s', n_pidx, sem, n_cls, n_iou, n_mask)
    assert torch.equal(n_pidx, pidx)
    assert torch.equal(n_cls, cls_s) and torch.equal(n_iou, iou_s) and torch.equal(n_mask, mask_s)
    assert len(preds_nat) == len(preds_op) > 0
    for a, c in zip(preds_nat, preds_op):
        assert a['label_id'] == c['label_id'] and a['conf'] == c['conf']
        assert a['pred_mask'] == c['pred_mask']
        assert type(a['label_id']) is type(c['label_id']) and type(a['conf']) is type(c['conf'])


def test_native_grouping_pieces_equal_operator_path():
    """the driver's intermediate products, one by one: proposals, voxel coordinates, voxel features,
    point -> voxel map, voxel offsets"""
    from softgroup_amd.model import native_scan as NS
    xyz, rgb, inst = synthetic.scene_s2(seed=5, n=60000, room_scale=0.6)
    b = _cuda(synthetic.make_batch(xyz, rgb, instance_labels=inst))
    model = synthetic.build_model(seed=1)
    g, v = model.grouping_cfg, model.instance_voxel_cfg
    with torch.no_grad():
        sem, off, out_feats = _backbone(model, b)
        pidx, poff = model.forward_grouping(sem, off, b['batch_idxs'], b['coords_float'], g)
        it, imap = model.clusters_voxelization(pidx, poff, out_feats, b['coords_float'], **v)
        _, seg_thr, _, cls32 = model._grouping_constants(sem.device)
        scores = sem.softmax(-1)
        cfg = NS.GroupingCfg(
            n_points=scores.size(0), n_sem_classes=scores.size(1), n_seg=cls32.numel(),
            seg_class=cls32.data_ptr(), seg_thr=seg_thr.data_ptr(), score_thr=g['score_thr'],
            min_npoint=model.test_cfg['min_npoint'], radius=g['radius'], batch_size=1,
            voxel_scale=v['scale'], voxel_shape=v['spatial_shape'], feat_channels=out_feats.size(1))
        r = NS.grouping(cfg, scores, off.contiguous(), b['coords_float'].contiguous(),
                        b['batch_idxs'].int().contiguous(), out_feats.contiguous())
    assert torch.equal(r['proposals_idx'], pidx) and torch.equal(r['proposals_offset'], poff.int())
    assert torch.equal(r['voxel_coords'], it.indices)
    assert torch.equal(r['voxel_feats'], it.features)
    assert torch.equal(r['point_to_voxel'], imap.int())
    counts = torch.bincount(it.indices[:, 0].long(), minlength=r['n_proposals'])
    assert torch.equal(r['voxel_offsets'][1:].long(), torch.cumsum(counts, 0))
    assert int(r['voxel_offsets'][0]) == 0


def test_native_proposals_on_a_two_scene_batch_equal_forward_grouping():
    """the training step's use of the driver (voxel_shape = 0: stop after the clustering) on a batch
    of two scenes: points of different scenes never share a proposal, values and order equal
    forward_grouping's (reference softgroup.py:411-480 with batch_idxs from collate_fn)"""
    from softgroup_amd.data import collate_device, make_item
    items = []
    for i in range(2):
        xyz, rgb, inst = synthetic.scene_s2(seed=11 + i, n=50000, room_scale=0.55)
        sem = np.where(inst >= 0, 2 + inst % 16, 0).astype(np.int64)
        items.append(make_item(xyz, rgb, 50, sem, inst, f's{i}'))
    b = collate_device(items)
    model = synthetic.build_model(seed=0)
    with torch.no_grad():
        sem, off, _ = _backbone(model, b)
        pidx, poff = model.forward_grouping(sem, off, b['batch_idxs'], b['coords_float'], model.grouping_cfg,
                                            batch_size=2)
        n_idx, n_off = model._native_proposals(sem, off, b['batch_idxs'], b['coords_float'], 2)
    assert pidx.shape[0] > 1000 and poff.numel() > 3
    assert torch.equal(n_idx, pidx) and torch.equal(n_off, poff.int())
    scene_of = b['batch_idxs'][pidx[:, 1].long()]
    first = scene_of[poff[:-1].long()]
    assert torch.equal(scene_of, first.repeat_interleave((poff[1:] - poff[:-1]).long()))


def test_native_grouping_without_proposals_falls_back_to_the_dummy_tensor():
    """no class passes the score threshold -> the driver reports nothing and forward_test takes the
    reference's dummy 2-voxel path (softgroup.py:664-673)"""
    xyz, rgb = synthetic.scene_s1(seed=0)
    b = _cuda(synthetic.make_batch(xyz, rgb, instance_labels=np.full(xyz.shape[0], -100, np.int64)))
    model = synthetic.build_model(seed=0, head_std=None)      # untrained head: flat class scores
    model.async_results = False
    with torch.no_grad():
        sem, off, out_feats = _backbone(model, b)
        r = model._native_grouping_and_refinement(sem, off, b['batch_idxs'], b['coords_float'], out_feats, 1)
        out_nat = model(b)
        model.use_native_scan = False
        out_op = model(b)
    assert r is None, 'flat class scores: nothing passes score_thr'
    assert len(out_nat['pred_instances']) == len(out_op['pred_instances'])


def test_forward_test_native_equals_operator_path_end_to_end():
    xyz, rgb, inst = synthetic.scene_s2(seed=7, n=80000, room_scale=0.7)
    b = _cuda(synthetic.make_batch(xyz, rgb, instance_labels=inst))
    model = synthetic.build_model(seed=0)
    model.async_results = False
    with torch.no_grad():
        out_one = dict(model(b))                    # sg_scan_forward: one C call
        model.use_scan_forward = False
        out_nat = dict(model(b))                    # staged: sg_scan_grouping / sg_scan_instances + torch heads
        model.use_native_scan = False
        out_op = dict(model(b))                     # operator surface
    assert len(out_nat['pred_instances']) == len(out_op['pred_instances']) > 0
    for a, c in zip(out_nat['pred_instances'], out_op['pred_instances']):
        assert a['label_id'] == c['label_id'] and a['conf'] == c['conf'] and a['pred_mask'] == c['pred_mask']
    np.testing.assert_array_equal(out_nat['semantic_preds'], out_op['semantic_preds'])
    # the one-call scan: same labels and RLE strings; its class / IoU heads are FMA chains (sg_linear_rows),
    # the staged path's a GEMM library -- confidences agree to rounding order
    assert len(out_one['pred_instances']) == len(out_op['pred_instances'])
    for a, c in zip(out_one['pred_instances'], out_op['pred_instances']):
        assert a['label_id'] == c['label_id'] and a['pred_mask'] == c['pred_mask']
        assert abs(float(a['conf']) - float(c['conf'])) <= 1e-6
    np.testing.assert_array_equal(out_one['semantic_preds'], out_op['semantic_preds'])


def test_panoptic_fusion_on_the_device_equals_the_reference_loop():
    """softgroup_kitti.yaml model section on a LiDAR-like sweep: sg_panoptic_fusion over the bit rows
    == SoftGroup.panoptic_fusion over the RLE strings (the reference's loop, softgroup.py:606-639,
    pinned to the reference's own forward in tests/test_ref_forward_golden.py)"""
    import copy
    xyz, intensity, inst = synthetic.scene_lidar(seed=3, n=60000)
    b = _cuda(synthetic.make_batch(xyz, intensity, scale=20, instance_labels=inst))
    model = synthetic.build_model(copy.deepcopy(synthetic.KITTI_MODEL_CFG), seed=0)
    model.async_results = False
    with torch.no_grad():
        out_nat = dict(model(b))
        # the same instances through the host loop
        sem_pred = torch.from_numpy(out_nat['semantic_preds']) if 'semantic_preds' in out_nat else None
        model.use_native_scan = False
        out_op = dict(model(b))
    assert out_nat['panoptic_preds'].dtype == np.uint32 == out_op['panoptic_preds'].dtype
    np.testing.assert_array_equal(out_nat['panoptic_preds'], out_op['panoptic_preds'])
    assert (out_nat['panoptic_preds'] >> 16).max() > 3, 'sweep must paste several instances'


def test_panoptic_fusion_dense_walk_equals_sparse_walk(monkeypatch):
    """the dense walk (rows above 1 M points, or SG_PANOPTIC_DENSE) had lost its
    `++next_id`.  Both walks over the same random, overlapping bit rows: identical words; the rows are
    also replayed on the host with the reference's loop (softgroup.py:606-639)."""
    from softgroup_amd import _lib as L
    lib = L.lib()
    rng = np.random.default_rng(7)
    n, n_inst = 5000, 40
    words = (n + 31) // 32
    masks = np.zeros((n_inst, n), bool)
    for k in range(n_inst):
        a = int(rng.integers(0, n - 400))
        masks[k, a:a + int(rng.integers(50, 400))] = True
        masks[k] &= rng.random(n) < 0.8
    bits = np.zeros((n_inst, words), np.uint32)
    for k in range(n_inst):
        idx = np.nonzero(masks[k])[0]
        np.bitwise_or.at(bits[k], idx >> 5, (np.uint32(1) << (idx & 31).astype(np.uint32)))
    order = rng.permutation(n_inst).astype(np.int32)
    labels = rng.integers(1, 9, n_inst).astype(np.int32)
    sem = rng.integers(0, 19, n).astype(np.int64)
    d = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    bits_d, order_d, lab_d, sem_d = d(bits.view(np.int32)), d(order), d(labels), d(sem)

    def run():
        out = torch.empty(n, dtype=torch.int32, device='cuda')
        ws = torch.empty(lib.sg_panoptic_fusion_workspace_bytes(n_inst, n), dtype=torch.uint8, device='cuda')
        L.check(lib.sg_panoptic_fusion(L.ptr(bits_d), n_inst, n, L.ptr(order_d), L.ptr(lab_d), L.ptr(sem_d), 10, 0.5,
                                       19, 11, L.ptr(out), L.ptr(ws), ws.numel(), L.stream()), 'sg_panoptic_fusion')
        return out.cpu().numpy().view(np.uint32)

    monkeypatch.delenv('SG_PANOPTIC_DENSE', raising=False)
    sparse = run()
    monkeypatch.setenv('SG_PANOPTIC_DENSE', '1')
    dense = run()
    # the reference's loop on the host
    want = sem.astype(np.uint32).copy()
    ids = np.zeros(n, np.uint32)
    taken = np.zeros(n, bool)
    nid = 1
    for k in order:
        m = masks[k]
        if (m & taken).sum() / (m.sum() + 1e-5) > 0.5:
            continue
        paste = m & ~taken
        want[paste], ids[paste] = labels[k] + 10, nid
        taken |= paste
        nid += 1
    exp = (want & 0xFFFF) | (ids << 16)
    exp[(want >= 11) & (ids == 0)] = 19
    assert nid > 5
    np.testing.assert_array_equal(sparse, exp)
    np.testing.assert_array_equal(dense, exp)


def test_panoptic_fusion_skips_overlapping_instances():
    """hand-made bit rows: the second instance overlaps the first by more than skip_iou and is
    skipped, the third is pasted on its free points only"""
    from softgroup_amd import _lib as L
    lib = L.lib()
    n = 200
    words = (n + 31) // 32
    masks = np.zeros((3, n), bool)
    masks[0, 10:60] = True
    masks[1, 20:70] = True          # 40 of 50 points taken -> 0.8 > 0.5: skipped
    masks[2, 50:120] = True         # 10 of 70 taken: pasted on 60..119
    bits = np.zeros((3, words), np.uint32)
    for k in range(3):
        for i in np.nonzero(masks[k])[0]:
            bits[k, i >> 5] |= np.uint32(1) << np.uint32(i & 31)
    sem = np.full(n, 2, np.int64)
    sem[150:] = 12                  # thing class nobody claims -> ignore label
    d = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    bits_d, order_d = d(bits.view(np.int32)), d(np.array([0, 1, 2], np.int32))
    lab_d, sem_d = d(np.array([1, 2, 3], np.int32)), d(sem)
    out = torch.empty(n, dtype=torch.int32, device='cuda')
    ws = torch.empty(lib.sg_panoptic_fusion_workspace_bytes(3, n), dtype=torch.uint8, device='cuda')
    L.check(lib.sg_panoptic_fusion(L.ptr(bits_d), 3, n, L.ptr(order_d), L.ptr(lab_d), L.ptr(sem_d), 10, 0.5, 19,
                                   11, L.ptr(out), L.ptr(ws), ws.numel(), L.stream()), 'sg_panoptic_fusion')
    got = out.cpu().numpy().view(np.uint32)
    want = sem.astype(np.uint32)
    ids = np.zeros(n, np.uint32)
    want[10:60], ids[10:60] = 1 + 10, 1
    want[60:120], ids[60:120] = 3 + 10, 2
    exp = (want & 0xFFFF) | (ids << 16)
    exp[(want >= 11) & (ids == 0)] = 19
    np.testing.assert_array_equal(got, exp)


@pytest.mark.parametrize('points,scale', [(150000, 40.0), (60000, 25.0)])
def test_softgroup_pp_grouping_in_c_equals_the_per_class_loop(points, scale):
    """sg_scan_grouping_pp (SoftGroup++: per-class pyramid level from the class size, level voxels with pooled
    coordinates / offsets, octree ball query at radius x level, clusters, inverse map; reference
    softgroup.py:433-466, functions.py:14-44) against the per-class loop over the operator surface
    (_grouping_per_class, which test_parity_at_size pins to the oracle): proposals, proposal voxels, pooled
    features and the scan's instances bit-identical.  150 k points x 40: the big class takes level 2."""
    import copy
    xyz, rgb, inst = synthetic.scene_s2(seed=1, n=points)
    xyz = (xyz * np.float32(scale)).astype(np.float32)
    batch = synthetic.make_batch(xyz, rgb, scale=3, instance_labels=inst)
    b = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    cfg = copy.deepcopy(synthetic.STPLS3D_PP_MODEL_CFG)
    model = synthetic.build_model(cfg, seed=0)
    model.async_results = False
    with torch.no_grad():
        vf = ops.voxelization(b['feats'], b['p2v_map'])
        x = spconv.SparseConvTensor(vf, b['voxel_coords'].int(), b['spatial_shape'], 1)
        sem, off, feats = model.forward_backbone(x, b['v2p_map'])
        bidx = b['batch_idxs'].int()
        # (a) the grouping stage alone
        pidx, poff = model.forward_grouping(sem, off, bidx, b['coords_float'], model.grouping_cfg, batch_size=1)
        assert poff.numel() - 1 >= 20
        from softgroup_amd.model import native_scan as NS
        g, v = model.grouping_cfg, model.instance_voxel_cfg
        _, seg_thr, _, cls32 = model._grouping_constants(sem.device)
        scores = sem.float().softmax(-1)
        base = NS.GroupingCfg(n_points=scores.size(0), n_sem_classes=scores.size(1), n_seg=cls32.numel(),
                              seg_class=cls32.data_ptr(), seg_thr=seg_thr.data_ptr(), score_thr=g['score_thr'],
                              min_npoint=model.test_cfg['min_npoint'], radius=g['radius'], batch_size=1,
                              voxel_scale=v['scale'], voxel_shape=v['spatial_shape'], feat_channels=feats.size(1))
        pp = NS.GroupingPPCfg(base=base, with_pyramid=1, with_octree=1, lvl_fusion=0, radius=float(g['radius']),
                              base_size=float(g['pyramid_base_size']))
        inst_t, inst_map = model.clusters_voxelization(pidx, poff, feats, b['coords_float'], **v)
        # with and without the deferred join (a class with a giant cluster lets the classes behind it run next
        # to its replay; at 150 k points the first grouped class holds one)
        import os
        for defer in ('1', '0'):
            os.environ['SG_PP_DEFER'] = defer
            try:
                r = NS.grouping(pp, scores, off.float().contiguous(), b['coords_float'].contiguous(),
                                bidx.contiguous(), feats.contiguous())
            finally:
                del os.environ['SG_PP_DEFER']
            assert r['deferred_classes'] == (1 if defer == '1' and points >= 150000 else 0), (defer, r['deferred_classes'])
            assert torch.equal(r['proposals_idx'], pidx) and torch.equal(r['proposals_offset'], poff), defer
            assert torch.equal(r['voxel_coords'], inst_t.indices) and torch.equal(r['voxel_feats'], inst_t.features)
            assert torch.equal(r['point_to_voxel'].long(), inst_map.long())
        # (b) the whole scan, C driver against the per-class loop
        assert model.use_native_grouping_pp
        out_c = dict(model(b))
        model.use_native_grouping_pp = False
        out_py = dict(model(b))
    assert len(out_c['pred_instances']) == len(out_py['pred_instances']) > 0
    for a, c in zip(out_c['pred_instances'], out_py['pred_instances']):
        assert a['label_id'] == c['label_id'] and a['pred_mask'] == c['pred_mask']
        assert abs(float(a['conf']) - float(c['conf'])) <= 1e-6
    np.testing.assert_array_equal(out_c['semantic_preds'], out_py['semantic_preds'])


@pytest.mark.parametrize('with_pyramid,with_octree,batch_size,lvl2', [(True, False, 1, False), (False, True, 1, False),
                                                                       (True, True, 2, False), (True, False, 2, True)])
def test_softgroup_pp_grouping_variants(with_pyramid, with_octree, batch_size, lvl2):
    """the branches of sg_scan_grouping_pp the STPLS3D configuration does not take: pyramid levels with the hashed-grid
    ball query, octree query without the pyramid, two scenes in a batch (the octree query ignores the batch index like
    the reference's, functions.py:14-44), a class above 100 000 points (level 2) -- against the per-class operator loop
    on synthetic scores, proposals and proposal voxels bit-identical"""
    import copy
    from softgroup_amd.model import native_scan as NS
    rng = np.random.default_rng(7 + batch_size)
    n_per = 130000 if lvl2 else 40000
    parts, bidx = [], []
    for b in range(batch_size):
        xyz, _, _ = synthetic.scene_s2(seed=31 + b, n=n_per)
        parts.append((xyz * np.float32(30)).astype(np.float32))
        bidx.append(np.full(n_per, b, np.int32))
    coords = torch.from_numpy(np.concatenate(parts)).cuda()
    batch_idxs = torch.from_numpy(np.concatenate(bidx)).cuda()
    N = coords.shape[0]
    cfg = copy.deepcopy(synthetic.STPLS3D_PP_MODEL_CFG)
    cfg['grouping_cfg'].update(with_pyramid=with_pyramid, with_octree=with_octree)
    model = synthetic.build_model(cfg, seed=0)
    n_cls = cfg['semantic_classes']
    # scores: a few big classes (one takes level 2 when lvl2), the rest small or empty
    logits = torch.full((N, n_cls), -8.0)
    cls_of = torch.from_numpy(rng.choice([1, 2, 4, 7, 9], size=N, p=[0.86 if lvl2 else 0.5, 0.05 if lvl2 else 0.2, 0.04 if lvl2 else 0.15, 0.04 if lvl2 else 0.1, 0.01 if lvl2 else 0.05]))
    logits[torch.arange(N), cls_of] = 4.0
    logits += torch.from_numpy(rng.normal(0, 0.5, (N, n_cls)).astype(np.float32))
    logits = logits.cuda()
    offs = torch.from_numpy(rng.normal(0, 0.05, (N, 3)).astype(np.float32)).cuda()
    feats = torch.from_numpy(rng.normal(0, 1, (N, 16)).astype(np.float32)).cuda()
    g, v = model.grouping_cfg, model.instance_voxel_cfg
    with torch.no_grad():
        pidx, poff = model.forward_grouping(logits, offs, batch_idxs, coords, g, batch_size=batch_size)
        assert poff.numel() - 1 >= 5
        _, seg_thr, _, cls32 = model._grouping_constants(logits.device)
        scores = logits.float().softmax(-1)
        base = NS.GroupingCfg(n_points=N, n_sem_classes=n_cls, n_seg=cls32.numel(), seg_class=cls32.data_ptr(),
                              seg_thr=seg_thr.data_ptr(), score_thr=g['score_thr'], min_npoint=model.test_cfg['min_npoint'],
                              radius=g['radius'], batch_size=batch_size, voxel_scale=v['scale'],
                              voxel_shape=v['spatial_shape'], feat_channels=16)
        pp = NS.GroupingPPCfg(base=base, with_pyramid=int(with_pyramid), with_octree=int(with_octree), lvl_fusion=0,
                              radius=float(g['radius']), base_size=float(g['pyramid_base_size']))
        r = NS.grouping(pp, scores, offs, coords, batch_idxs, feats)
        assert torch.equal(r['proposals_offset'], poff) and torch.equal(r['proposals_idx'], pidx)
        inst_t, inst_map = model.clusters_voxelization(pidx, poff, feats, coords, **v)
        assert torch.equal(r['voxel_coords'], inst_t.indices) and torch.equal(r['voxel_feats'], inst_t.features)
    if lvl2:
        assert int((cls_of == 1).sum()) > 100000      # the big class really took level 2
