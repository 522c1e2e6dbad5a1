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
