"""TEST INFRASTRUCTURE.  Builds, from the reference sources where they lie under
/root/reference/softgroup/ops/src (nothing is copied into the repo):

  oracle/_ref/sg_ref_ops.so      the reference's own CPU ops (voxelize_idx, bfs_cluster,
                                 build_and_export_octree), g++ + two shim headers (oracle/ref_shims)
  oracle/_ref/sg_ref_gpu_ops.so  the reference's own CUDA kernels (cuda.cu: voxelize fp/bp, ball
                                 query, octree ball query, sec_mean/min/max, ROI avg pool fp/bp,
                                 mask IoU x2, mask label), hipcc --offload-arch=gfx950 + a
                                 cuda*->hip* shim (oracle/ref_shims_gpu), C entry points in
                                 oracle/ref_gpu_tu.hip, Python front end oracle/ref_gpu.py

  oracle/_ref/pysrc/             the reference's own PYTHON (softgroup/{model,util,evaluation,data}
                                 and tools/test.py) byte-compiled from the sources where they lie
                                 into sourceless .pyc files -- a build output like the .so files:
                                 no reference source text enters the repository.  It lets the GPU
                                 box (which has no /root/reference) import the reference's SoftGroup
                                 class, dataset, evaluator and test loop on top of the HIP operators
                                 (tests/test_dropin_gpu.py); oracle/facade.py falls back to it.

Outputs go only to oracle/_ref/ (git-ignored, but shipped to the GPU box).  Skips silently when
/root/reference is absent (GPU box: the prebuilt files are used)."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = '/root/reference/softgroup/ops/src'
OUT_DIR = os.path.join(HERE, '_ref')
OUT = os.path.join(OUT_DIR, 'sg_ref_ops.so')
OUT_GPU = os.path.join(OUT_DIR, 'sg_ref_gpu_ops.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
REF_ROOT = '/root/reference'
PY_OUT = os.path.join(OUT_DIR, 'pysrc')
PY_PACKAGES = ('softgroup/model', 'softgroup/util', 'softgroup/evaluation', 'softgroup/data')
PY_FILES = ('tools/test.py', 'tools/train.py')


def build_py(force=False):
    """byte-compile the reference's Python packages (sourceless import layout: <pkg>/<mod>.pyc).
    The .pyc magic number is the interpreter's; the GPU box runs the same image."""
    import py_compile
    if not os.path.isdir(os.path.join(REF_ROOT, 'softgroup')):
        return PY_OUT if os.path.isdir(PY_OUT) else None
    jobs = []
    for pkg in PY_PACKAGES:
        d = os.path.join(REF_ROOT, pkg)
        jobs += [(os.path.join(d, f), os.path.join(PY_OUT, pkg, f + 'c'))
                 for f in sorted(os.listdir(d)) if f.endswith('.py')]
    jobs += [(os.path.join(REF_ROOT, f), os.path.join(PY_OUT, f + 'c')) for f in PY_FILES]
    for src, dst in jobs:
        if not force and os.path.exists(dst) and os.path.getmtime(dst) > os.path.getmtime(src):
            continue
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        py_compile.compile(src, cfile=dst, dfile=os.path.relpath(src, REF_ROOT), doraise=True)
    with open(os.path.join(PY_OUT, 'MAGIC'), 'wb') as f:
        import importlib.util
        f.write(importlib.util.MAGIC_NUMBER)
    return PY_OUT


def build_gpu(force=False):
    """hipcc build of the reference's CUDA kernels (cross-compiles without a GPU).  Default
    floating-point contraction (hipcc's -ffp-contract=fast-honor-pragmas is the analogue of nvcc's
    default -fmad=true the reference is built with, setup.py:15-23).  -fno-slp-vectorize keeps the
    contraction the one of LLVM's generic DAG combiner, which nvcc's NVVM back end shares
    (mul dy,dy; fma dx,dx; fma dz,dz for the d2 of the ball queries); with SLP on, hipcc packs
    dx*dx and dz*dz into one v_pk_mul_f32 first and a different product gets rounded -- an
    AMDGPU-only artefact that no CUDA build of the reference can have."""
    if not os.path.isdir(REF_SRC):
        return OUT_GPU if os.path.exists(OUT_GPU) else None
    src = os.path.join(HERE, 'ref_gpu_tu.hip')
    shim = os.path.join(HERE, 'ref_shims_gpu')
    deps = [src, os.path.join(shim, 'sg_cuda_on_hip.h')]
    if (not force and os.path.exists(OUT_GPU)
            and all(os.path.getmtime(OUT_GPU) > os.path.getmtime(d) for d in deps)):
        return OUT_GPU
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = [HIPCC, '-O3', '-std=c++17', '-fPIC', '-shared', '--offload-arch=gfx950', '-w', '-fno-slp-vectorize',
           f'-I{shim}', f'-I{REF_SRC}', src, '-o', OUT_GPU]
    subprocess.check_call(cmd)
    return OUT_GPU


def build(force=False):
    build_gpu(force)
    build_py(force)
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
