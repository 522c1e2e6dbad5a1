"""Generates the golden batch of the REFERENCE'S OWN dataset + ``collate_fn`` (authoring container
only: needs /root/reference).

Two small synthetic scans are written as ScanNet-style ``*_inst_nostuff.pth`` files (what
dataset/scannetv2/prepare_data_inst.py produces: xyz float32, rgb float32 in [-1, 1], semantic and
instance labels as float64 with -100 = unlabelled; the second scan has a gap in its instance ids).
The reference's ``ScanNetDataset`` (softgroup/data/scannetv2.py over data/custom.py, imported from
where it lies) loads them at test time -- ``transform_test`` incl. its fixed 0.35 pi rotation,
``getCroppedInstLabel``, ``getInstanceInfo`` -- and its ``collate_fn`` (custom.py:196-256; the
voxel index through the reference's C++ ``voxelization_idx`` compiled into oracle/_ref) builds the
batch dict.  Stored: the raw scans and every entry of the items and of the batch.

tests/test_data_golden.py checks ``softgroup_amd.data.scan_item`` (CPU) and tests/test_data_gpu.py
``collate_device`` (GPU) against it, value for value.

A second file, ref_collate_variants.npz, holds the items of the other dataset classes run the same
way -- ``S3DISDataset`` at test time with ``x4_split`` (four interleaved sub-clouds, its own
``collate_fn`` branch, data/s3dis.py:46-115), ``STPLS3DDataset`` (class shift 1, voxel scale 3) and
``KITTIDataset`` (``.bin`` / ``.label`` files, label words, learning map, ranked instance ids,
data/kitti.py) -- plus the S3DIS test-time batch.

Usage:  python tests/golden/make_ref_collate.py
"""
import importlib
import logging
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import facade  # noqa: E402
from softgroup_amd import synthetic  # noqa: E402

SCENES = [dict(seed=21, n=3000, room_scale=0.14), dict(seed=22, n=2400, room_scale=0.12)]
VOXEL_CFG = dict(scale=50, spatial_shape=[128, 512], max_npoint=250000, min_npoint=5000)


def raw_scan(i):
    xyz, rgb, inst = synthetic.scene_s2(**SCENES[i])
    inst = inst.astype(np.float64)
    if i == 1:                       # a gap in the ids (as after cropping): id 3 is missing
        inst[inst == 3] = -100
    sem = np.where(inst >= 0, 2 + inst % 18, 0).astype(np.float64)
    sem[::97] = -100                 # some unlabelled points
    return xyz, rgb, sem, inst


def main():
    os.environ['TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD'] = '1'     # the reference calls torch.load(file) on numpy tuples
    facade.import_reference()
    mod = importlib.import_module('softgroup.data.scannetv2')
    rec = {}
    with tempfile.TemporaryDirectory() as root:
        os.makedirs(os.path.join(root, 'val'))
        for i in range(len(SCENES)):
            xyz, rgb, sem, inst = raw_scan(i)
            torch.save((xyz, rgb, sem, inst), os.path.join(root, 'val', f'scene{i:04d}_00_inst_nostuff.pth'))
            rec.update({f'raw{i}_xyz': xyz, f'raw{i}_rgb': rgb, f'raw{i}_sem': sem, f'raw{i}_inst': inst})
        ds = mod.ScanNetDataset(root, 'val', '_inst_nostuff.pth', voxel_cfg=facade.NS(VOXEL_CFG), training=False,
                                with_label=True, logger=logging.getLogger('ref'))
        items = [ds[i] for i in range(len(ds))]
        names = ('scan_id', 'coord', 'coord_float', 'feat', 'semantic_label', 'instance_label', 'inst_num',
                 'inst_pointnum', 'inst_cls', 'pt_offset_label')
        for i, it in enumerate(items):
            for k, v in zip(names, it):
                # (a copy: collate_fn shifts the instance ids of the items IN PLACE, custom.py:216)
                rec[f'item{i}_{k}'] = np.array(v.numpy() if isinstance(v, torch.Tensor) else v, copy=True)
        batch = ds.collate_fn(items)
    for k, v in batch.items():
        rec[f'batch_{k}'] = np.asarray(v.numpy() if isinstance(v, torch.Tensor) else v)
    path = os.path.join(HERE, 'ref_collate.npz')
    np.savez_compressed(path, **rec)
    print({k: (v.shape, str(v.dtype)) for k, v in rec.items() if k.startswith('batch_') or k.startswith('item1_')})
    print(os.path.getsize(path) // 1024, 'KiB')
    variants()


KITTI_YAML = dict(
    split=dict(train=[0], valid=[8], test=[11]),
    # (a subset of the SemanticKITTI table: raw label -> learning id; 1..8 things, 9..19 stuff)
    learning_map={0: 0, 1: 0, 10: 1, 11: 2, 15: 3, 18: 4, 30: 6, 40: 9, 48: 11, 50: 13, 70: 15, 71: 16, 80: 18},
    learning_map_inv={0: 0, 1: 10, 2: 11, 3: 15, 4: 18, 6: 30, 9: 40, 11: 48, 13: 50, 15: 70, 16: 71, 18: 80})


def kitti_raw():
    """a sweep-like cloud: xyz + intensity and 32-bit label words (instance << 16 | class)"""
    xyz, rgb, inst = synthetic.scene_s2(seed=25, n=2000, room_scale=0.12)
    things, stuff = [10, 11, 15, 18, 30], [40, 48, 50, 70, 71, 80, 0, 1]
    cls = np.where(inst >= 0, np.array(things)[np.clip(inst, 0, None) % 5], np.array(stuff)[np.arange(len(inst)) % 8])
    word = (np.where(inst >= 0, (inst * 7 + 3), 0).astype(np.int64) << 16 | cls).astype(np.int32)
    return np.concatenate([xyz * 8, rgb[:, :1]], 1).astype(np.float32), word


def variants():
    """S3DIS (x4_split, test), STPLS3D and SemanticKITTI: the items of the reference's dataset classes, and
    the S3DIS test-time batch of its own collate_fn"""
    import yaml
    rec = {}
    names = ('scan_id', 'coord', 'coord_float', 'feat', 'semantic_label', 'instance_label', 'inst_num',
             'inst_pointnum', 'inst_cls', 'pt_offset_label')
    log = logging.getLogger('ref')
    cfg = facade.NS(VOXEL_CFG)

    def put(tag, item):
        for k, v in zip(names, item):
            rec[f'{tag}_item_{k}'] = np.array(v.numpy() if isinstance(v, torch.Tensor) else v, copy=True)

    with tempfile.TemporaryDirectory() as root:
        # ---- S3DIS: 6-tuple files, 13 classes, instance classes unshifted, four sub-clouds at test time
        xyz, rgb, inst = synthetic.scene_s2(seed=23, n=2402, room_scale=0.12)
        inst = inst.astype(np.float64)
        inst[inst == 5] = -100
        sem = np.where(inst >= 0, inst % 13, 1).astype(np.float64)
        torch.save((xyz, rgb, sem, inst, None, None), os.path.join(root, 'Area_5_office_1_inst_nostuff.pth'))
        rec.update(s3dis_raw_xyz=xyz, s3dis_raw_rgb=rgb, s3dis_raw_sem=sem, s3dis_raw_inst=inst)
        mod = importlib.import_module('softgroup.data.s3dis')
        ds = mod.S3DISDataset(x4_split=True, data_root=root, prefix='Area_5', suffix='_inst_nostuff.pth',
                              voxel_cfg=cfg, training=False, with_label=True, logger=log)
        item = ds[0]
        put('s3dis', item)
        for k, v in ds.collate_fn([item]).items():
            rec[f's3dis_batch_{k}'] = np.asarray(v.numpy() if isinstance(v, torch.Tensor) else v)
        # ---- STPLS3D: class 0 is not an instance class (shift 1), voxel scale 3
        os.makedirs(os.path.join(root, 'val'))
        xyz, rgb, inst = synthetic.scene_s2(seed=24, n=1800, room_scale=0.12)
        xyz = (xyz * 20).astype(np.float32)
        inst = inst.astype(np.float64)
        sem = np.where(inst >= 0, 1 + inst % 14, 0).astype(np.float64)
        torch.save((xyz, rgb, sem, inst), os.path.join(root, 'val', '5_points_GTv3_00_inst_nostuff.pth'))
        rec.update(stpls3d_raw_xyz=xyz, stpls3d_raw_rgb=rgb, stpls3d_raw_sem=sem, stpls3d_raw_inst=inst)
        mod = importlib.import_module('softgroup.data.stpls3d')
        ds = mod.STPLS3DDataset(root, 'val', '_inst_nostuff.pth',
                                voxel_cfg=facade.NS(dict(VOXEL_CFG, scale=3)), training=False, with_label=True,
                                logger=log)
        put('stpls3d', ds[0])
        # ---- SemanticKITTI: .bin / .label files, label words, learning map, ranked instance ids
        yaml.safe_dump(KITTI_YAML, open(os.path.join(root, 'semantic-kitti.yaml'), 'w'))
        vd, ld = os.path.join(root, 'sequences', '08', 'velodyne'), os.path.join(root, 'sequences', '08', 'labels')
        os.makedirs(vd)
        os.makedirs(ld)
        data, word = kitti_raw()
        data.tofile(os.path.join(vd, '000000.bin'))
        word.tofile(os.path.join(ld, '000000.label'))
        rec.update(kitti_raw_data=data, kitti_raw_word=word,
                   kitti_learning_map=np.array(sorted(KITTI_YAML['learning_map'].items()), np.int64))
        mod = importlib.import_module('softgroup.data.kitti')
        ds = mod.KITTIDataset(root, 'val', '.bin', voxel_cfg=facade.NS(dict(VOXEL_CFG, scale=20)), training=False,
                              with_label=True, logger=log)
        put('kitti', ds[0])
    path = os.path.join(HERE, 'ref_collate_variants.npz')
    np.savez_compressed(path, **rec)
    print({k: (v.shape, str(v.dtype)) for k, v in rec.items() if '_batch_' in k or k.startswith('kitti_item')})
    print(os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
