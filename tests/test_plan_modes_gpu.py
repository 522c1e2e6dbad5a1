"""The developer A/B knobs of round 5's locality experiments (SG_UNET_MORTON, SG_PLAN_ORDER, SG_PLAN_SB:
internal Morton row order of the executor, spatially local tile plans; profiles/r05_conv_locality.txt) are
read once per process, so each configuration runs in a subprocess: the U-Net executor's features under the
knobs must equal the module path's (the operator-by-operator forward the other GPU tests pin to the oracle)
within the conv tolerance, on a scene whose levels span several super-blocks."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SNIPPET = r'''
import sys, torch
sys.path.insert(0, %r)
from softgroup_amd import ops, synthetic
import softgroup_amd.spconv.pytorch as spconv
xyz, rgb, inst = synthetic.scene_s2(seed=4, n=60000, room_scale=0.63)
b = synthetic.make_batch(xyz, rgb, instance_labels=inst)
b = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
model = synthetic.build_model(seed=0)
with torch.no_grad():
    vf = ops.voxelization(torch.cat((b['feats'], b['coords_float']), 1), b['p2v_map'])
    x = spconv.SparseConvTensor(vf, b['voxel_coords'].int(), b['spatial_shape'], 1)
    a = model._unet_features(x)
    a2 = model._unet_features(x)
    model.use_executor = False
    m = model._unet_features(x)
scale = float(m.abs().max())
print('RESULT', x.features.shape[0], float((a - m).abs().max()) / scale, bool(torch.equal(a, a2)))
''' % ROOT


@pytest.mark.parametrize('env', [
    dict(SG_UNET_MORTON='1', SG_UNET_MORTON_MIN='1000'),
    dict(SG_UNET_MORTON='1', SG_UNET_MORTON_MIN='1000', SG_PLAN_ORDER='1', SG_PLAN_SB='4096'),
    dict(SG_UNET_MORTON='1', SG_UNET_MORTON_MIN='1000', SG_PLAN_ORDER='2'),
    dict(SG_PLAN_ORDER='1'),
], ids=['morton', 'morton+local4096', 'morton+local_sbmajor', 'local_only'])
def test_executor_under_locality_knobs_equals_module_path(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, '-c', SNIPPET], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('RESULT')][-1].split()
    rows, rel, repeat = int(line[1]), float(line[2]), line[3] == 'True'
    assert rows > 20000
    assert rel <= 1e-4, f'executor vs module path: {rel:.3e} of the feature scale'      # conv tolerance
    assert repeat, 'two forwards of the same input must be bit-identical'
