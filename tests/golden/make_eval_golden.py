"""Runs the REFERENCE's own evaluator (/root/reference/softgroup/evaluation/instance_eval.py,
imported from where it lies; authoring container only) on the deterministic inputs of
eval_cases.py and stores its averages -> tests/golden/eval_golden.json.
numpy >= 1.24 dropped np.float / np.bool, which the reference still uses: aliased for the run."""
import importlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import eval_cases  # noqa: E402
from oracle import facade  # noqa: E402


def main():
    np.float, np.bool = float, bool
    import types
    ply = types.ModuleType('plyfile')          # instance_eval_util imports it for file export only
    ply.PlyData = ply.PlyElement = object
    sys.modules.setdefault('plyfile', ply)
    facade.import_reference()
    ref = importlib.import_module('softgroup.evaluation.instance_eval')
    out = {}
    for name, kw in (('class_aware', dict(use_label=True)), ('class_agnostic', dict(use_label=False)),
                     ('min_npoint_30', dict(use_label=True, min_npoint=30))):
        ev = ref.ScanNetEval(list(eval_cases.CLASSES), **kw)
        pl, gl = eval_cases.cases()
        avgs = ev.evaluate(pl, gl)
        out[name] = json.loads(json.dumps(avgs, default=float))
    json.dump(out, open(os.path.join(HERE, 'eval_golden.json'), 'w'), indent=1, sort_keys=True)
    print({k: (v['all_ap'], v['all_ap_50%'], v['all_ap_25%']) for k, v in out.items()})


if __name__ == '__main__':
    main()
