"""Device-side collate (softgroup_amd.data.collate_device) against the reference's
CPU collate semantics: same items in, same batch dict out -- every key, dtype and value -- with
the voxel index built by the HIP voxelization_idx instead of the CPU op (data/custom.py:239)."""
import numpy as np
import pytest
import torch

from softgroup_amd import ops, synthetic
from softgroup_amd.data import collate_device, make_item

pytestmark = pytest.mark.gpu


def _cpu_collate(items, min_spatial=128):
    """collate_fn as the reference runs it (data/custom.py:196-256), on the CPU"""
    coords, total, bid = [], 0, 0
    cf, ft, sem, ins, pn, cl, off = [], [], [], [], [], [], []
    for (sid, coord, coord_float, feat, s, i, inum, ipn, icl, po) in items:
        i = i.clone()
        i[np.where(i != -100)] += total
        total += inum
        coords.append(torch.cat([coord.new_full((coord.size(0), 1), bid), coord], 1))
        cf.append(coord_float); ft.append(feat); sem.append(s); ins.append(i)
        pn.extend(ipn); cl.extend(icl); off.append(po)
        bid += 1
    coords = torch.cat(coords, 0)
    vc, v2p, p2v = ops.voxelization_idx(coords, bid)           # CPU tensors -> host op
    return dict(coords=coords, batch_idxs=coords[:, 0].int(), voxel_coords=vc, p2v_map=p2v, v2p_map=v2p,
                coords_float=torch.cat(cf).float(), feats=torch.cat(ft), semantic_labels=torch.cat(sem).long(),
                instance_labels=torch.cat(ins).long(), instance_pointnum=torch.tensor(pn, dtype=torch.int),
                instance_cls=torch.tensor(cl, dtype=torch.long), pt_offset_labels=torch.cat(off).float(),
                spatial_shape=np.clip(coords.max(0)[0][1:].numpy() + 1, min_spatial, None), batch_size=bid)


def test_collate_device_equals_cpu_collate_two_scenes():
    items = []
    for seed, n in ((3, 20000), (4, 12000)):
        xyz, rgb, inst = synthetic.scene_s2(seed=seed, n=n, room_scale=0.4)
        items.append(make_item(xyz, rgb, 50, None, inst, f'scene{seed}'))
    ref = _cpu_collate(items)
    got = collate_device(items)
    assert got['scan_ids'] == ['scene3', 'scene4'] and got['batch_size'] == 2
    assert np.array_equal(got['spatial_shape'], ref['spatial_shape'])
    for k, v in ref.items():
        if isinstance(v, torch.Tensor):
            g = got[k]
            assert g.is_cuda and g.dtype == v.dtype and g.shape == v.shape, k
            assert torch.equal(g.cpu(), v), k
    # second call reuses the pinned staging buffers: results must not alias the first call's
    again = collate_device(items[:1])
    assert again['batch_size'] == 1 and torch.equal(got['coords'].cpu(), ref['coords'])


def test_make_item_matches_make_batch_and_runs_the_model():
    xyz, rgb, inst = synthetic.scene_s2(seed=5, n=20000, room_scale=0.37)
    b_dev = collate_device([make_item(xyz, rgb, 50, None, inst, 'synthetic_0000')])
    b_cpu = synthetic.make_batch(xyz, rgb, instance_labels=inst)
    for k in ('coords', 'voxel_coords', 'p2v_map', 'v2p_map', 'coords_float', 'feats', 'instance_labels',
              'semantic_labels', 'instance_pointnum', 'instance_cls', 'pt_offset_labels'):
        assert torch.equal(b_dev[k].cpu(), b_cpu[k]), k
    model = synthetic.build_model(seed=0)
    with torch.no_grad():
        a, b = model(b_dev), model(b_cpu)
    assert len(a['pred_instances']) == len(b['pred_instances']) > 0
    assert all(x['pred_mask'] == y['pred_mask'] for x, y in zip(a['pred_instances'], b['pred_instances']))


def test_collate_device_equals_the_reference_dataset_and_collate_fn():
    """The reference's own ``ScanNetDataset.__getitem__`` + ``collate_fn`` (run as written on two
    synthetic scans, tests/golden/make_ref_collate.py -> ref_collate.npz; second scan with a gap in
    its instance ids, unlabelled points, float64 labels as in the prepared files) against
    ``data.scan_item`` + ``collate_device``: every key of the batch dict, dtype and value."""
    from test_data_golden import GOLD, items_from_golden
    g = np.load(GOLD)
    batch = collate_device(items_from_golden(g))
    torch.cuda.synchronize()
    keys = [k[len('batch_'):] for k in g.files if k.startswith('batch_')]
    assert set(keys) == set(batch.keys())
    for k in keys:
        ref, got = g['batch_' + k], batch[k]
        if isinstance(got, torch.Tensor):
            assert got.is_cuda, k
            got = got.cpu().numpy()
        got = np.asarray(got)
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        if ref.dtype.kind in 'fiu':
            assert got.dtype == ref.dtype, (k, got.dtype, ref.dtype)
        assert np.array_equal(got, ref), k


def test_collate_x4_device_equals_the_reference_s3dis_test_collate():
    """S3DIS at test time (x4_split): the reference's ``S3DISDataset.collate_fn`` branch for one scan
    whose item holds four interleaved sub-clouds (golden ref_collate_variants.npz) against
    ``scan_item(x4_split=True)`` + ``collate_x4_device`` -- every key, dtype and value, quirks
    included (no ``coords``, ``batch_idxs`` all zero, instance lists with a leading dimension)."""
    from softgroup_amd.data import collate_x4_device
    from test_data_golden import VARIANTS, variant_item
    g = np.load(VARIANTS)
    batch = collate_x4_device([variant_item(g, 's3dis')])
    torch.cuda.synchronize()
    keys = [k[len('s3dis_batch_'):] for k in g.files if k.startswith('s3dis_batch_')]
    assert set(keys) == set(batch.keys())
    for k in keys:
        ref, got = g['s3dis_batch_' + k], batch[k]
        if isinstance(got, torch.Tensor):
            assert got.is_cuda, k
            got = got.cpu().numpy()
        got = np.asarray(got)
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        if ref.dtype.kind in 'fiu':
            assert got.dtype == ref.dtype, (k, got.dtype, ref.dtype)
        assert np.array_equal(got, ref), k


@pytest.mark.parametrize('workers', [1, 2])
def test_prefetch_device_hands_over_the_same_batches_and_results(workers):
    """data.prefetch_device: the next scan's collate on a loader thread / stream (the reference's DataLoader
    workers ahead of tools/test.py:145) -- every batch equal to collate_device's key by key, the scans' results
    equal to the ones computed from batches collated in line, an exception of the loader reaches the consumer"""
    from softgroup_amd.data import prefetch_device
    items = []
    for i in range(4):
        xyz, rgb, inst = synthetic.scene_s2(seed=20 + i, n=20000 + 3000 * i, room_scale=0.4)
        items.append(make_item(xyz, rgb, 50, None, inst, f's{i}'))
    model = synthetic.build_model(seed=0)
    model.async_results = False
    with torch.no_grad():
        want = [collate_device([it]) for it in items]
        ref = [dict(model(b)) for b in want]
        got = []
        for k, b in enumerate(prefetch_device([[it] for it in items], workers=workers, depth=2)):
            for key, v in want[k].items():
                if isinstance(v, torch.Tensor):
                    assert torch.equal(b[key], v), key
                elif isinstance(v, np.ndarray):
                    assert np.array_equal(b[key], v), key
                else:
                    assert b[key] == v, key
            got.append(dict(model(b)))
    assert len(got) == 4
    for a, c in zip(got, ref):
        assert len(a['pred_instances']) == len(c['pred_instances'])
        for x, y in zip(a['pred_instances'], c['pred_instances']):
            assert x['label_id'] == y['label_id'] and x['conf'] == y['conf'] and x['pred_mask'] == y['pred_mask']
        np.testing.assert_array_equal(a['semantic_preds'], c['semantic_preds'])

    def broken():
        yield [items[0]]
        raise RuntimeError('loader failed')

    with pytest.raises(RuntimeError, match='loader failed'):
        for _ in prefetch_device(broken(), workers=workers):
            pass
