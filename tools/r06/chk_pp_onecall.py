import copy, sys, os
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from softgroup_amd import synthetic
from softgroup_amd.model.softgroup import _cfg
xyz, rgb, inst = synthetic.scene_s2(seed=1, n=150000)
xyz = (xyz * np.float32(40)).astype(np.float32)
batch = synthetic.make_batch(xyz, rgb, scale=3, instance_labels=inst)
batch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
model = synthetic.build_model(copy.deepcopy(synthetic.STPLS3D_PP_MODEL_CFG), seed=0)
with torch.no_grad():
    r = dict(model(batch))
    sf = model.__dict__.get('_scan_forward')
    print('scan_forward object', sf is not None)
    if sf is not None:
        t = model.test_cfg
        print('usable', sf.usable(model, _cfg(t, 'eval_tasks'), _cfg(t, 'lvl_fusion', False), _cfg(t, 'x4_split', False)))
        print('last stage', getattr(sf, 'last', None) and sf.last.stage, 'deferred', getattr(sf, 'last', None) and sf.last.grouping.deferred_classes)
    print(len(r['pred_instances']))
