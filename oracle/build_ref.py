"""TEST INFRASTRUCTURE.  Builds, from the reference sources where they lie under
/root/reference/softgroup/ops/src (nothing is copied into the repo):

  oracle/_ref/sg_ref_ops.so      the reference's own CPU ops (voxelize_idx, bfs_cluster,
                                 build_and_export_octree), g++ + two shim headers (oracle/ref_shims)
  oracle/_ref/sg_ref_gpu_ops.so  the reference's own CUDA kernels (cuda.cu: voxelize fp/bp, ball
                                 query, octree ball query, sec_mean/min/max, ROI avg pool fp/bp,
                                 mask IoU x2, mask label), hipcc --offload-arch=gfx950 + a
                                 cuda*->hip* shim (oracle/ref_shims_gpu), C entry points in
                                 oracle/ref_gpu_tu.hip, Python front end oracle/ref_gpu.py

Outputs go only to oracle/_ref/ (git-ignored, but shipped to the GPU box).  Skips silently when
/root/reference is absent (GPU box: the prebuilt .so files are used)."""
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
