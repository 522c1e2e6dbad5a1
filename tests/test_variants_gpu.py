"""Parity of the other BASELINE configs at small size (they are parity cases, not bench lines):
SoftGroup++/STPLS3D (octree ball query + pyramid levels + lvl_fusion, channels 16),
S3DIS (x4_split, sem2ins classes) and SemanticKITTI (1 input channel, no coords, panoptic fusion).
HIP-hosted model vs the CPU restatement of the reference model, stage by stage on identical inputs
(float features 1e-4, integer products bit-exact), same weights."""
import copy

import numpy as np
import pytest
import torch

import softgroup_amd.spconv.pytorch as spconv
from oracle.model import OracleSoftGroup, SparseT
from softgroup_amd import ops, synthetic

pytestmark = pytest.mark.gpu
TOL = dict(atol=1e-4, rtol=1e-4)


def n(t):
    return t.detach().cpu().numpy()


def run_case(cfg, batch, get_level=None, min_props=1):
    model = synthetic.build_model(cfg, seed=0)
    ora = OracleSoftGroup(model.state_dict(), cfg)
    if get_level is not None:                       # exercise pyramid level > 1 at test size
        model.get_level = get_level
        ora.get_level = get_level
    tc = cfg['test_cfg']
    lvl, x4 = tc.get('lvl_fusion', False), tc['x4_split']
    with torch.no_grad():
        b = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
        feats = torch.cat((b['feats'], b['coords_float']), 1) if cfg.get('with_coords', True) else b['feats']
        vf = ops.voxelization(feats.contiguous(), b['p2v_map'])
        x = spconv.SparseConvTensor(vf, b['voxel_coords'].int(), b['spatial_shape'], b['batch_size'])
        sem, off, of = model.forward_backbone(x, b['v2p_map'], x4_split=x4, lvl_fusion=lvl)
        cf, bi = b['coords_float'], b['batch_idxs']
        if x4:
            cf = model.merge_4_parts(cf)
        if lvl:
            bi = x.indices[:, 0].int()
            cf = ops.voxelization(cf.contiguous(), b['p2v_map'])
        pidx, poff = model.forward_grouping(sem, off, bi, cf, model.grouping_cfg, lvl_fusion=lvl)
        inst, imap = model.clusters_voxelization(pidx, poff, of, cf, **model.instance_voxel_cfg)
        _, cls_s, iou_s, mask_s = model.forward_instance(inst, imap)
        preds = model.get_instances('s', pidx, sem, cls_s, iou_s, mask_s, v2p_map=b['v2p_map'],
                                    lvl_fusion=lvl)
        full = model(batch)

    # (1) backbone + heads
    osem, ooff, ofeat = ora.point_wise(batch, x4, lvl)
    np.testing.assert_allclose(n(of), ofeat, **TOL)
    np.testing.assert_allclose(n(sem), osem, **TOL)                  # (logits are O(10..100) here; measured 3e-5)
    np.testing.assert_allclose(n(off), ooff, **TOL)
    # (2) grouping on the same scores
    rp, ro = ora.grouping(n(sem), n(off), n(bi), n(cf), lvl)
    assert len(ro) - 1 >= min_props, f'only {len(ro) - 1} proposals: scene does not exercise grouping'
    assert np.array_equal(n(pidx), rp) and np.array_equal(n(poff), ro)
    # (3) proposal voxelisation
    oinst, oimap = ora.clusters_voxelization(rp, ro, n(of), n(cf))
    assert np.array_equal(n(inst.indices), oinst.indices) and np.array_equal(n(imap), oimap)
    np.testing.assert_allclose(n(inst.features), oinst.features, **TOL)
    # (4) tiny U-Net + heads
    same = SparseT(n(inst.features), oinst.indices, oinst.spatial_shape, oinst.batch_size)
    ocls, oiou, omask = ora.instance_heads(same, oimap)
    np.testing.assert_allclose(n(cls_s), ocls, **TOL)
    np.testing.assert_allclose(n(iou_s), oiou, **TOL)
    np.testing.assert_allclose(n(mask_s), omask, **TOL)
    # (5) instances / RLE
    ref = ora.get_instances('s', rp, n(sem), n(cls_s), n(iou_s), n(mask_s), batch['v2p_map'], lvl)
    assert len(preds) == len(ref)
    for a, r in zip(preds, ref):
        assert a['label_id'] == r['label_id'] and a['pred_mask'] == r['pred_mask']
        assert abs(float(a['conf']) - float(r['conf'])) < 1e-6
    return model, ora, full, preds, dict(sem=n(sem))


def test_softgroup_pp_stpls3d_octree_pyramid():
    """softgroup++_stpls3d.yaml: channels 16 (Cout not a multiple of 32), octree ball query (no
    batch separation, leaf order), pyramid voxels; level forced to 2 for classes > 1500 points."""
    cfg = copy.deepcopy(synthetic.STPLS3D_PP_MODEL_CFG)
    xyz, rgb, inst = synthetic.scene_s2(seed=7, n=8000, room_scale=0.24)
    xyz = (xyz * np.float32(22.5)).astype(np.float32)
    batch = synthetic.make_batch(xyz, rgb, scale=3, instance_labels=inst)
    lvl2 = lambda num: 2 if num > 1500 else 1  # noqa: E731
    run_case(cfg, batch, get_level=lvl2)
    # test-time lvl_fusion: grouping and heads run on voxels, masks expand through v2p_map
    cfg['test_cfg']['lvl_fusion'] = True
    run_case(cfg, batch, get_level=lvl2)


def test_s3dis_x4_split_and_sem2ins():
    cfg = copy.deepcopy(synthetic.S3DIS_MODEL_CFG)
    xyz, rgb, inst = synthetic.scene_s2(seed=11, n=40000, room_scale=0.5)
    batch = synthetic.make_batch(xyz, rgb, instance_labels=inst, x4_split=True)
    model, ora, full, preds, _ = run_case(cfg, batch)
    # sem2ins classes 0,1 come first, one instance each with conf 1 (softgroup.py:556-561)
    assert [p['label_id'] for p in preds[:2]] == [1, 2] and preds[0]['conf'] == 1.0
    assert full['semantic_preds'].shape[0] == xyz.shape[0]


def test_kitti_panoptic():
    cfg = copy.deepcopy(synthetic.KITTI_MODEL_CFG)
    xyz, rgb, inst = synthetic.scene_s2(seed=13, n=30000, room_scale=0.45)
    xyz = (xyz * np.float32(2.5)).astype(np.float32)
    batch = synthetic.make_batch(xyz, rgb[:, :1].copy(), scale=20, instance_labels=inst)
    model, ora, full, preds, aux = run_case(cfg, batch)
    assert 'panoptic_preds' in full and full['panoptic_preds'].dtype == np.uint32
    ref = ora.panoptic_fusion(aux['sem'].argmax(1), preds)
    got = model.panoptic_fusion(aux['sem'].argmax(1), preds)
    assert np.array_equal(got, ref)
