"""The test-time item of the data path against the REFERENCE'S OWN dataset class
(tests/golden/ref_collate.npz, made by tests/golden/make_ref_collate.py from
softgroup/data/scannetv2.py + custom.py run as written).  CPU part: ``data.scan_item``;
the batch built from these items by ``collate_device`` is checked in tests/test_data_gpu.py."""
import os

import numpy as np
import torch

from softgroup_amd import data

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_collate.npz')
NAMES = ('scan_id', 'coord', 'coord_float', 'feat', 'semantic_label', 'instance_label', 'inst_num',
         'inst_pointnum', 'inst_cls', 'pt_offset_label')


def items_from_golden(g):
    """our items for the raw scans stored in the golden file"""
    out = []
    for i in range(2):
        out.append(data.scan_item(g[f'raw{i}_xyz'], g[f'raw{i}_rgb'], g[f'raw{i}_sem'], g[f'raw{i}_inst'],
                                  scale=50, scan_id=str(g[f'item{i}_scan_id'])))
    return out


def test_scan_item_equals_the_reference_getitem():
    g = np.load(GOLD)
    for i, item in enumerate(items_from_golden(g)):
        for k, v in zip(NAMES, item):
            ref = g[f'item{i}_{k}']
            got = np.asarray(v.numpy() if isinstance(v, torch.Tensor) else v)
            assert got.shape == ref.shape, (i, k, got.shape, ref.shape)
            if got.dtype.kind in 'fiu':
                assert got.dtype == ref.dtype, (i, k, got.dtype, ref.dtype)
            assert np.array_equal(got, ref), (i, k)
    # the second scan has a gap in its ids: the last id moved into it (getCroppedInstLabel)
    raw, lab = g['raw1_inst'], g['item1_instance_label']
    assert raw.max() == 11 and lab.max() == 10 and not (raw == 3).any() and (lab == 3).sum() == (raw == 11).sum()
    # and the fixed test-time rotation is in the coordinates
    assert not np.allclose(g['item0_coord_float'], g['raw0_xyz'], atol=1e-3)


VARIANTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_collate_variants.npz')


def variant_item(g, tag):
    """our item for the raw scan of a dataset variant stored in the golden file"""
    sid = str(g[f'{tag}_item_scan_id'])
    if tag == 's3dis':
        return data.scan_item(g['s3dis_raw_xyz'], g['s3dis_raw_rgb'], g['s3dis_raw_sem'], g['s3dis_raw_inst'],
                              scale=50, scan_id=sid, cls_shift=0, x4_split=True)
    if tag == 'stpls3d':
        return data.scan_item(g['stpls3d_raw_xyz'], g['stpls3d_raw_rgb'], g['stpls3d_raw_sem'],
                              g['stpls3d_raw_inst'], scale=3, scan_id=sid, cls_shift=1)
    raw = g['kitti_raw_data']
    sem, inst = data.kitti_labels(g['kitti_raw_word'], {int(k): int(v) for k, v in g['kitti_learning_map']})
    return data.scan_item(raw[:, :3], raw[:, 3:], sem, inst, scale=20, scan_id=sid, cls_shift=11, relabel='rank')


def test_scan_item_equals_the_reference_getitem_of_the_other_datasets():
    """S3DISDataset at test time with x4_split (four interleaved sub-clouds, each shifted to its own
    minimum, piece number in column 0), STPLS3DDataset (class shift 1) and KITTIDataset (label words
    through the learning map, stuff points without instance, ids ranked): every field of the item the
    reference's classes return for the same raw scan."""
    g = np.load(VARIANTS)
    for tag in ('s3dis', 'stpls3d', 'kitti'):
        for k, v in zip(NAMES, variant_item(g, tag)):
            ref = g[f'{tag}_item_{k}']
            got = np.asarray(v.numpy() if isinstance(v, torch.Tensor) else v)
            assert got.shape == ref.shape, (tag, k, got.shape, ref.shape)
            if got.dtype.kind in 'fiu':
                assert got.dtype == ref.dtype, (tag, k, got.dtype, ref.dtype)
            assert np.array_equal(got, ref), (tag, k)
    assert g['s3dis_item_coord'].shape[1] == 4 and set(np.unique(g['s3dis_item_coord'][:, 0])) == {0, 1, 2, 3}
    assert g['kitti_item_inst_num'] == 12 and (g['kitti_item_inst_cls'] >= 0).all() and (g['kitti_item_inst_cls'] < 8).all()
