"""Builds softgroup_amd/lib/libsoftgroup_hip.so with hipcc for gfx950 (cross-compiles without a
GPU).  In-tree output so the library travels with the repo snapshot to the GPU box."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
OBJDIR = os.path.join(LIBDIR, 'obj')
LIB = os.path.join(LIBDIR, 'libsoftgroup_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')

# (source, extra flags).  -ffp-contract=off wherever results must be bit-exact with the oracle.
SOURCES = [
    ('core.hip', ['-ffp-contract=off']),
    ('seg_ops.hip', ['-ffp-contract=off']),
    ('voxelize_idx.hip', ['-ffp-contract=off']),
    ('ballquery.hip', ['-ffp-contract=off']),
    ('octree.hip', ['-ffp-contract=off']),
    ('bfs.hip', ['-ffp-contract=off']),
    ('spconv_rulebook.hip', ['-ffp-contract=off']),
    # (atomic optimizer off: it would wait for the unit-ticket atomic right where it is issued)
    ('spconv_conv.hip', ['-mllvm', '-amdgpu-mfma-vgpr-form', '-mllvm', '-amdgpu-atomic-optimizer-strategy=None'] + os.environ.get('SG_CONV_EXTRA_FLAGS', '').split()),
    ('spconv_train.hip', ['-mllvm', '-amdgpu-mfma-vgpr-form']),
    ('unet_exec.hip', ['-ffp-contract=off']),
    ('unet_train.hip', ['-ffp-contract=off']),
    ('instances.hip', ['-ffp-contract=off']),
    ('scan_exec.hip', ['-ffp-contract=off']),
    ('scan_forward.hip', ['-ffp-contract=off']),
    ('heads.hip', ['-ffp-contract=off']),
    ('eval_ops.hip', ['-ffp-contract=off']),
    ('host_ops.cpp', ['-ffp-contract=off']),
]
COMMON = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-Wall', '-Wno-unused-function',
          '-Wno-unused-result', '-Wno-unused-value']


def _deps_mtime():
    m = 0
    for d in (CSRC, os.path.join(os.path.dirname(HERE), 'include')):
        for f in os.listdir(d):
            if f.endswith('.h'):
                m = max(m, os.path.getmtime(os.path.join(d, f)))
    return m


def _compile(src, flags, hdr_mtime, force):
    path = os.path.join(CSRC, src)
    if not os.path.exists(path):
        return None
    obj = os.path.join(OBJDIR, src + '.o')
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(path)
            and os.path.getmtime(obj) > hdr_mtime):
        return obj
    cmd = [HIPCC] + COMMON + flags
    if src.endswith('.cpp'):
        cmd += ['-x', 'hip']
    cmd += ['-c', path, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError(f'hipcc failed on {src}')
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    hdr = _deps_mtime()
    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(lambda sf: _compile(sf[0], sf[1], hdr, force), SOURCES))
    objs = [o for o in objs if o]
    if (force or not os.path.exists(LIB)
            or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs)):
        cmd = [HIPCC, '-shared', '-fPIC', '--offload-arch=gfx950', '-o', LIB] + objs
        subprocess.check_call(cmd)
        if verbose:
            print('linked', LIB)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
