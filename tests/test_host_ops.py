"""Host (CPU) entry points of the library -- the two operators the reference itself runs on the
CPU -- against the golden vectors produced by the reference's own C++."""
import numpy as np
import torch

from softgroup_amd import _lib as L
from softgroup_amd import ops


def test_voxelization_idx_cpu_matches_reference(golden):
    g = golden('voxelize_idx')
    for i in range(int(g['vox_ncases'])):
        for mode in (4, 3):
            p = f'vox{i}_m{mode}_'
            oc, im, om = ops.voxelization_idx(torch.from_numpy(g[p + 'coords']), int(g[p + 'batch']),
                                              mode)
            assert not oc.is_cuda and oc.dtype == torch.int64 and im.dtype == torch.int32
            assert np.array_equal(oc.numpy(), g[p + 'out_coords'])
            assert np.array_equal(im.numpy(), g[p + 'input_map'])
            assert np.array_equal(om.numpy(), g[p + 'output_map'])


def test_voxelization_idx_cpu_edge_cases():
    oc, im, om = ops.voxelization_idx(torch.zeros((0, 4), dtype=torch.int64), 1)
    assert oc.shape == (0, 4) and im.shape == (0,) and om.shape == (0, 2)
    c = torch.tensor([[0, -5, 7, 1 << 20]] * 3)
    oc, im, om = ops.voxelization_idx(c, 1)
    assert om.tolist() == [[3, 0, 1, 2]] and oc.tolist() == [[0, -5, 7, 1 << 20]]
    # modes 1 (first) / 2 (last), voxelize.cpp:134-149
    c = torch.tensor([[0, 1, 1, 1], [0, 2, 2, 2], [0, 1, 1, 1]])
    assert ops.voxelization_idx(c, 1, 1)[2].tolist() == [[1, 0], [1, 1]]
    assert ops.voxelization_idx(c, 1, 2)[2].tolist() == [[1, 2], [1, 1]]


def test_octree_build_host_matches_reference(golden):
    g = golden('octree')
    lib = L.lib()
    for k in range(int(g['oct_ncases'])):
        p = f'oct{k}_'
        pts = torch.from_numpy(g[p + 'points'])
        n = pts.shape[0]
        boxes = torch.zeros((585, 6))
        pt_inds = torch.zeros(n, dtype=torch.int32)
        psl = torch.zeros((512, 2), dtype=torch.int32)
        L.check(lib.sg_octree_build_host(L.ptr(pts), L.ptr(torch.from_numpy(g[p + 'xyzwhl'])), n, 3,
                                         L.ptr(boxes), L.ptr(pt_inds), L.ptr(psl)))
        assert np.array_equal(boxes.numpy(), g[p + 'boxes'])
        assert np.array_equal(pt_inds.numpy(), g[p + 'pt_inds'])
        assert np.array_equal(psl.numpy(), g[p + 'pt_start_len'])


def test_collate_staging_helpers_match_numpy():
    """the plain-C staging copies of the device-side collate (sg_host_copy_2d, sg_host_cast_f64_f32,
    sg_host_fill_i64_strided, sg_host_colmax_i64: what data/custom.py:196-256 does with torch.cat / max on the
    CPU) against numpy, including empty inputs and the generic column count."""
    lib = L.lib()
    rng = np.random.default_rng(3)
    src = rng.integers(-50, 700, (1001, 3)).astype(np.int64)
    dst = np.zeros((1001, 4), np.int64)
    L.check(lib.sg_host_fill_i64_strided(dst.ctypes.data, 4, 1001, 7), 'fill')
    L.check(lib.sg_host_copy_2d(dst.ctypes.data + 8, 32, src.ctypes.data, 24, 1001, 24), 'copy')
    assert (dst[:, 0] == 7).all() and np.array_equal(dst[:, 1:], src)
    d = rng.normal(0, 1e3, 777)
    f = np.empty(777, np.float32)
    L.check(lib.sg_host_cast_f64_f32(f.ctypes.data, d.ctypes.data, 777), 'cast')
    assert np.array_equal(f, d.astype(np.float32))
    for cols in (1, 3, 4, 8):
        m = rng.integers(-1000, 1000, (513, cols)).astype(np.int64)
        out = np.empty(cols, np.int64)
        L.check(lib.sg_host_colmax_i64(m.ctypes.data, 513, cols, out.ctypes.data), 'colmax')
        assert np.array_equal(out, m.max(0)), cols
    out = np.empty(3, np.int64)
    L.check(lib.sg_host_colmax_i64(None, 0, 3, out.ctypes.data), 'colmax empty')
    assert (out == np.iinfo(np.int64).min).all()
    assert lib.sg_host_colmax_i64(src.ctypes.data, 10, 9, out.ctypes.data) != 0      # more than 8 columns: refused
