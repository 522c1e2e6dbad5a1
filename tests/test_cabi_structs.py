"""The descriptor structs of include/softgroup_hip.h against their ctypes mirrors in the Python
bindings: sizes and the offset of every field, taken from the header itself by compiling a probe
with gcc (the header is plain C).  A field added on one side only would otherwise shift every
pointer behind it silently."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'softgroup_hip.h')


def _mirrors():
    from softgroup_amd.model import native_scan as NS
    from softgroup_amd.spconv import unet_exec as UE
    from softgroup_amd.spconv import unet_train as UT
    from softgroup_amd.model import softgroup as SGM
    from softgroup_amd.model import scan_forward as SF
    return {
        'sg_unet_block': UE._Block, 'sg_unet_level': UE._Level, 'sg_unet_desc': UE._Desc,
        'sg_train_bn': UT._TBn, 'sg_train_conv': UT._TConv, 'sg_unet_train_block': UT._TBlock,
        'sg_unet_train_level': UT._TLevel, 'sg_unet_train_desc': UT._TDesc,
        'sg_grouping_cfg': NS.GroupingCfg, 'sg_grouping_pp_cfg': NS.GroupingPPCfg, 'sg_grouping_result': NS.GroupingResult,
        'sg_instances_cfg': NS.InstancesCfg, 'sg_instances_result': NS.InstancesResult,
        'sg_mlp2': SGM._Mlp2, 'sg_linear': SF.Linear, 'sg_scan_dense_item': SF.DenseItem,
        'sg_scan_desc': SF.ScanDesc, 'sg_scan_input': SF.ScanInput, 'sg_scan_result': SF.ScanResult,
    }


def _header_fields(name):
    """field names of `typedef struct <name> { ... } <name>;` in declaration order"""
    txt = re.sub(r'/\*.*?\*/', '', open(HEADER).read(), flags=re.S)
    m = re.search(r'typedef struct %s\s*\{(.*?)\}\s*%s\s*;' % (name, name), txt, flags=re.S)
    assert m, f'{name} not found in the header'
    fields = []
    for decl in m.group(1).split(';'):
        decl = decl.strip()
        if not decl:
            continue
        # "const float *a, *b" / "int x, y" / "sg_train_bn bn1, bn2" -> names
        for part in decl.split(','):
            nm = re.findall(r'([A-Za-z_][A-Za-z0-9_]*)\s*(?:\[[^\]]*\])?\s*$', part.strip())
            assert nm, (name, decl)
            fields.append(nm[0])
    return fields


@pytest.fixture(scope='module')
def probe():
    """{struct: (size, {field: offset})} as gcc lays the header's structs out"""
    names = list(_mirrors())
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', 'int main(void) {']
    for n in names:
        lines.append(f'  printf("S {n} %zu\\n", sizeof({n}));')
        for f in _header_fields(n):
            lines.append(f'  printf("F {n} {f} %zu\\n", offsetof({n}, {f}));')
    lines += ['  return 0;', '}']
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, 'probe.c'), os.path.join(d, 'probe')
        open(src, 'w').write('\n'.join(lines))
        subprocess.run(['gcc', '-std=c11', '-o', exe, src], check=True, capture_output=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    res = {}
    for ln in out.splitlines():
        p = ln.split()
        if p[0] == 'S':
            res[p[1]] = (int(p[2]), {})
        else:
            res[p[1]][1][p[2]] = int(p[3])
    return res


def test_every_descriptor_has_the_same_layout_on_both_sides(probe):
    for name, mirror in _mirrors().items():
        size, offsets = probe[name]
        assert C.sizeof(mirror) == size, (name, C.sizeof(mirror), size)
        py_fields = [f[0] for f in mirror._fields_]
        assert py_fields == list(offsets), (name, py_fields, list(offsets))
        for f in py_fields:
            assert getattr(mirror, f).offset == offsets[f], (name, f)
