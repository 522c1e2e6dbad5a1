"""Generates the end-to-end golden vectors of the REFERENCE'S OWN Python model (run in the
authoring container only: needs /root/reference).

For each of the four BASELINE model configs the reference's ``SoftGroup`` class
(/root/reference/softgroup/model/softgroup.py, imported from where it lies) is built from the
``model:`` section of the reference's own YAML, loaded with our seeded synthetic weights, and its
``forward_test`` is executed AS WRITTEN on a small synthetic scene -- on the CPU, with its two native
dependencies (spconv, softgroup.ops) replaced by the C-oracle-backed stand-ins of oracle/facade.py.

Outputs (committed; tests only read them):
  tests/golden/ref_configs.json         the ``model:`` section of every YAML under
                                        /root/reference/configs (so the GPU box can check the
                                        config contract without the reference tree)
  tests/golden/ref_forward_<case>.npz   scene recipe + checksum, semantic_preds, offset_preds,
                                        pred_instances (label_id, conf, RLE counts), panoptic_preds

They pin (i) oracle/model.py -- our restatement of that control flow -- exactly
(tests/test_ref_forward_golden.py, CPU) and (ii) the HIP-hosted model end to end
(tests/test_dropin_gpu.py, GPU).

Usage:  python tests/golden/make_ref_forward.py
"""
import copy
import glob
import json
import os
import sys

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import facade  # noqa: E402
from softgroup_amd import synthetic  # noqa: E402

REF_CFG = '/root/reference/configs'


def lvl2(num_points):
    """pyramid level 2 already for classes > 1500 points, so the small scene exercises it"""
    return 2 if num_points > 1500 else 1


# case -> (yaml, scene recipe).  Scenes are regenerated from the recipe by the tests
# (numpy Generator streams are stable); a checksum of the coordinates guards against drift.
CASES = {
    'scannet': dict(yaml='softgroup/softgroup_scannet.yaml',
                    scene=dict(seed=7, n=16000, room_scale=0.33), xyz_scale=1.0, vox_scale=50),
    'stpls3d_pp': dict(yaml='softgroup++/softgroup++_stpls3d.yaml',
                       scene=dict(seed=7, n=8000, room_scale=0.24), xyz_scale=22.5, vox_scale=3,
                       force_lvl2=True),
    's3dis': dict(yaml='softgroup/softgroup_s3dis_fold5.yaml',
                  scene=dict(seed=11, n=24000, room_scale=0.4), xyz_scale=1.0, vox_scale=50,
                  x4_split=True),
    'kitti': dict(yaml='softgroup/softgroup_kitti.yaml',
                  scene=dict(seed=13, n=30000, room_scale=0.45), xyz_scale=2.5, vox_scale=20,
                  one_channel=True),
}


def make_case_batch(case):
    """the deterministic scene + batch dict of a case (also used by the tests)"""
    c = CASES[case]
    xyz, rgb, inst = synthetic.scene_s2(**c['scene'])
    xyz = (xyz * np.float32(c['xyz_scale'])).astype(np.float32)
    if c.get('one_channel'):
        rgb = rgb[:, :1].copy()
    return synthetic.make_batch(xyz, rgb, scale=c['vox_scale'], instance_labels=inst,
                                x4_split=c.get('x4_split', False)), xyz


def main():
    # ---- every model section of the reference's configs
    configs = {}
    for path in sorted(glob.glob(os.path.join(REF_CFG, '*', '*.yaml'))):
        configs[os.path.relpath(path, REF_CFG)] = yaml.safe_load(open(path))['model']
    json.dump(configs, open(os.path.join(HERE, 'ref_configs.json'), 'w'), indent=1, sort_keys=True)
    print('ref_configs.json:', len(configs), 'configs')

    for case, c in CASES.items():
        cfg = copy.deepcopy(configs[c['yaml']])
        batch, xyz = make_case_batch(case)
        sd = synthetic.build_model(cfg, seed=0, device='cpu').state_dict()
        ref, mod = facade.reference_model(cfg, sd)
        if c.get('force_lvl2'):
            ref.get_level = lvl2
        with facade.cpu_only(mod), torch.no_grad():
            out = ref(batch)
        preds = out.get('pred_instances', [])
        rec = dict(
            yaml=np.array(c['yaml']), recipe=np.array(json.dumps(c)),
            xyz_checksum=np.float64(np.abs(xyz.astype(np.float64)).sum()),
            n_points=np.int64(xyz.shape[0]),
            label_id=np.array([int(p['label_id']) for p in preds], np.int64),
            conf=np.array([float(p['conf']) for p in preds], np.float32),
            rle_length=np.array([int(p['pred_mask']['length']) for p in preds], np.int64),
            rle_counts=np.array([p['pred_mask']['counts'] for p in preds]),
        )
        if 'semantic_preds' in out:
            rec['semantic_preds'] = out['semantic_preds'].astype(np.int16)
            rec['offset_preds'] = out['offset_preds'].astype(np.float32)
        if 'gt_instances' in out:
            rec['gt_instances'] = np.asarray(out['gt_instances']).astype(np.int64)
        if 'panoptic_preds' in out:
            rec['panoptic_preds'] = out['panoptic_preds'].astype(np.uint32)
        path = os.path.join(HERE, f'ref_forward_{case}.npz')
        np.savez_compressed(path, **rec)
        print(case, 'points', xyz.shape[0], 'instances', len(preds), 'keys', sorted(out),
              os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
