"""TEST INFRASTRUCTURE.  Builds oracle/_ref/sg_ref_ops.so: the reference's own
CPU ops (voxelize_idx, bfs_cluster, build_and_export_octree) compiled unmodified
from /root/reference/softgroup/ops/src with two shim headers (oracle/ref_shims).
No reference source is copied into the repo; outputs go only to oracle/_ref/
(git-ignored, but shipped to the GPU box).  Skips silently when /root/reference
is absent (GPU box: the prebuilt .so is used)."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = '/root/reference/softgroup/ops/src'
OUT_DIR = os.path.join(HERE, '_ref')
OUT = os.path.join(OUT_DIR, 'sg_ref_ops.so')


def build(force=False):
    if not os.path.isdir(REF_SRC):
        return OUT if os.path.exists(OUT) else None
    src = os.path.join(HERE, 'ref_tu.cpp')
    if (not force and os.path.exists(OUT)
            and os.path.getmtime(OUT) > os.path.getmtime(src)):
        return OUT
    import torch
    from torch.utils import cpp_extension
    os.makedirs(OUT_DIR, exist_ok=True)
    tdir = os.path.dirname(torch.__file__)
    incs = [os.path.join(HERE, 'ref_shims'), REF_SRC] + cpp_extension.include_paths() + \
        [sysconfig.get_paths()['include']]
    cmd = ['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-w',
           '-DTORCH_EXTENSION_NAME=sg_ref_ops', '-DTORCH_API_INCLUDE_EXTENSION_H',
           f'-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}']
    cmd += [f'-I{p}' for p in incs]
    cmd += [src, '-o', OUT, f'-L{tdir}/lib', '-ltorch', '-ltorch_cpu', '-lc10', '-ltorch_python',
            f'-Wl,-rpath,{tdir}/lib']
    subprocess.check_call(cmd)
    return OUT


def load():
    """Import the built reference ops module (or None if unavailable)."""
    import importlib.util
    import torch  # noqa: F401  (libtorch must be loaded first)
    if not os.path.exists(OUT):
        return None
    spec = importlib.util.spec_from_file_location('sg_ref_ops', OUT)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
