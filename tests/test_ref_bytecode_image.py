"""The GPU box has no /root/reference: the reference's Python travels there as sourceless bytecode
(oracle/_ref/pysrc, built by oracle/build_ref.py from the sources where they lie).  This checks, on
the CPU and in a fresh interpreter that is told the reference tree does not exist, that the image
is complete: the model module, the dataset / evaluation packages and tools/test.py import from it
and the reference's SoftGroup class constructs from its own YAML model section."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PYSRC = os.path.join(ROOT, 'oracle', '_ref', 'pysrc')

CODE = r'''
import importlib, json, sys
sys.path.insert(0, %(root)r)
from oracle import facade
assert facade.ref_root() == %(pysrc)r, facade.ref_root()
mod = facade.import_reference()
assert mod.__file__.endswith('.pyc')
import scipy.interpolate, scipy.ndimage, numpy.ma            # (before any np.bool alias)
data = importlib.import_module('softgroup.data')
ev = importlib.import_module('softgroup.evaluation')
tool = facade.import_reference_tool('test')
cfg = json.load(open(%(cfgs)r))['softgroup/softgroup_scannet.yaml']
c = {k: (facade.Munch.fromDict(v) if isinstance(v, dict) else v) for k, v in cfg.items()}
m = mod.SoftGroup(**c)
print('OK', sum(p.numel() for p in m.parameters()), tool.main.__name__, data.build_dataset.__name__,
      ev.ScanNetEval.__name__)
'''


@pytest.mark.skipif(not os.path.isdir(PYSRC), reason='oracle/_ref/pysrc not built (__graft_entry__.build())')
def test_reference_python_imports_from_the_bytecode_image():
    code = CODE % dict(root=ROOT, pysrc=PYSRC, cfgs=os.path.join(ROOT, 'tests', 'golden', 'ref_configs.json'))
    env = dict(os.environ, SG_REF_ROOT='/nonexistent')
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith('OK 30839600 main build_dataset ScanNetEval'), r.stdout
