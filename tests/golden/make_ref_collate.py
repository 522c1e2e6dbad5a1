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

tests/test_data_gpu.py checks ``softgroup_amd.data.scan_item`` (CPU part) and ``collate_device``
(GPU) against it, value for value.

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


if __name__ == '__main__':
    main()
