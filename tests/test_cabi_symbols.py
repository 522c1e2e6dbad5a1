"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol that
include/softgroup_hip.h declares, and the ctypes table covers exactly that set."""
import ctypes
import os
import re

from softgroup_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, 'include', 'softgroup_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return set(re.findall(r'\b(sg_[a-z0-9_]+)\s*\(', txt))


def test_library_exports_every_declared_symbol():
    syms = _header_symbols()
    assert len(syms) > 30
    raw = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in sorted(syms) if not hasattr(raw, s)]
    assert not missing, missing


def test_ctypes_table_matches_header():
    assert set(_lib.SIGNATURES) == _header_symbols()


def test_version_and_error_text():
    lib = _lib.lib()
    assert lib.sg_version() >= 1
    # bad argument -> negative status + readable message, no crash
    rc = lib.sg_voxelize_idx_host(None, -1, 4, 4, None, None, None)
    assert rc < 0 and b'sg_voxelize_idx_host' in lib.sg_last_error()


def test_reference_gpu_oracle_library_exports_its_entry_points():
    """oracle/_ref/sg_ref_gpu_ops.so (the reference's CUDA kernels built with hipcc, test-only):
    loads without a GPU and exports every sgref_* entry point oracle/ref_gpu.py calls."""
    import pytest
    from oracle import ref_gpu
    if not ref_gpu.available():
        pytest.skip('oracle/_ref/sg_ref_gpu_ops.so not built (/root/reference absent at build time)')
    raw = ctypes.CDLL(ref_gpu.SO)
    src = open(os.path.join(ROOT, 'oracle', 'ref_gpu.py')).read()
    used = set(re.findall(r"\b(sgref_[a-z_]+)", src))
    assert len(used) >= 14
    missing = [s for s in sorted(used) if not hasattr(raw, s)]
    assert not missing, missing
