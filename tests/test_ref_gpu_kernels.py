"""Pins the CUDA-only kernel families against the REFERENCE'S OWN kernels.

oracle/_ref/sg_ref_gpu_ops.so = /root/reference/softgroup/ops/src/cuda.cu compiled unmodified with
hipcc for gfx950 (oracle/build_ref.py; built in the CPU container, shipped to the GPU box like the
other built .so files; never imported by softgroup_amd/).  For every family two comparisons on
the same seeded inputs:

  (i)  oracle/sg_oracle.c  == reference kernels      (pins the CPU restatement)
  (ii) HIP kernels (C ABI) == reference kernels      (parity proper)

Bar: bit-exact (integer products, order-preserving fp32 sums, min/max, f64-quotient IoU).  The
reference ball queries place their lists with an atomic cursor (bfs_cluster.cu:53,
octree_ball_query.cu:113), so lists are compared per point, not by start offset."""
import numpy as np
import pytest
import torch

import oracle
from oracle import ref_gpu
from softgroup_amd import ops

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not ref_gpu.available(),
                                 reason='oracle/_ref/sg_ref_gpu_ops.so not built (needs /root/reference at build time)')]
DEV = 'cuda'


def t(a, dtype=None):
    x = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return x if dtype is None else x.to(dtype)


def _lists(idx, sl):
    """per-point neighbour lists from a CSR with arbitrary start offsets"""
    idx, sl = np.asarray(idx), np.asarray(sl)
    return [idx[s:s + l] for s, l in sl]


def _same_lists(idx_a, sl_a, idx_b, sl_b):
    if not np.array_equal(np.asarray(sl_a)[:, 1], np.asarray(sl_b)[:, 1]):
        return False
    return all(np.array_equal(x, y) for x, y in zip(_lists(idx_a, sl_a), _lists(idx_b, sl_b)))


def _blob_cloud(rng, n_blobs, per_blob, n_noise, sigma=0.03):
    ctr = rng.random((n_blobs, 3)) * np.array([6, 5, 2.7])
    pts = [ctr[i] + rng.normal(0, sigma, (per_blob, 3)) for i in range(n_blobs)]
    pts.append(rng.random((n_noise, 3)) * np.array([6, 5, 2.7]))
    xyz = np.concatenate(pts).astype(np.float32)
    return xyz[rng.permutation(len(xyz))]


def _segments(rng, nP, lo, hi):
    lens = rng.integers(lo, hi, nP)
    return np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)


# ----------------------------------------------------------------------------- voxelize fp / bp
@pytest.mark.parametrize('C', [6, 32, 3, 33])
def test_voxelize_fp_bp(C):
    rng = np.random.default_rng(100 + C)
    n = 40000
    c = rng.integers(0, 24, (n, 4)).astype(np.int64)
    c[:, 0] = 0
    _, _, om = oracle.voxelization_idx(c, 1)
    feats = rng.standard_normal((n, C)).astype(np.float32)
    g = rng.standard_normal((om.shape[0], C)).astype(np.float32)
    for mode in (4, 3):
        ref = ref_gpu.voxelization(t(feats), t(om), mode).cpu().numpy()
        assert np.array_equal(oracle.voxelization(feats, om, mode), ref)                  # (i)
        f = t(feats).requires_grad_(True)
        out = ops.voxelization(f, t(om), mode)
        assert np.array_equal(out.detach().cpu().numpy(), ref)                            # (ii)
        ref_bp = ref_gpu.voxelization_bp(t(g), t(om), n, mode).cpu().numpy()
        assert np.array_equal(oracle.voxelization_bp(g, om, n, mode), ref_bp)
        out.backward(t(g))
        assert np.array_equal(f.grad.cpu().numpy(), ref_bp)


# ----------------------------------------------------------------------------- ball query
def _ballquery_case(xyz, bi, radius):
    n = xyz.shape[0]
    B = int(bi.max()) + 1
    bo = np.concatenate([[0], np.cumsum(np.bincount(bi, minlength=B))]).astype(np.int32)
    ridx, rsl = ref_gpu.ballquery_batch_p(t(xyz), t(bi), t(bo), radius, 300)
    ridx, rsl = ridx.cpu().numpy(), rsl.cpu().numpy()
    oidx, osl = oracle.ballquery_batch_p(xyz, bi, bo, radius, 300)
    assert _same_lists(oidx, osl, ridx, rsl)                                              # (i)
    idx, sl = ops.ballquery_batch_p(t(xyz), t(bi), t(bo), radius, 300)
    assert _same_lists(idx.cpu().numpy(), sl.cpu().numpy(), ridx, rsl)                    # (ii)
    assert int(rsl[:, 1].sum()) == len(ridx)
    return rsl


def test_ballquery_blobs_two_batches():
    rng = np.random.default_rng(2)
    xyz = _blob_cloud(rng, 12, 300, 2000)
    bi = np.sort(rng.integers(0, 2, len(xyz))).astype(np.int32)
    _ballquery_case(xyz, bi, 0.04)


def test_ballquery_lattice_on_the_radius():
    """points exactly r apart and 1e-4 off: strict '<' on d2 and the contraction of the reference's
    d2 expression decide membership here"""
    rng = np.random.default_rng(5)
    g = np.stack(np.meshgrid(*[np.arange(-6, 6)] * 3, indexing='ij'), -1).reshape(-1, 3)
    xyz = (g * 0.05).astype(np.float32)
    xyz = np.concatenate([xyz, xyz + np.float32(1e-4), (rng.random((500, 3)) - 0.5).astype(np.float32)])
    _ballquery_case(xyz, np.zeros(len(xyz), np.int32), 0.05)


def test_ballquery_random_near_radius_pairs():
    """many pairs within a few ulp of the radius"""
    rng = np.random.default_rng(6)
    a = (rng.random((3000, 3)) * 2).astype(np.float32)
    d = rng.standard_normal((3000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = np.float32(0.04)
    b = (a.astype(np.float64) + d * (float(r) * (1 + rng.uniform(-3e-7, 3e-7, (3000, 1))))).astype(np.float32)
    xyz = np.concatenate([a, b])
    _ballquery_case(xyz, np.zeros(len(xyz), np.int32), float(r))


def test_ballquery_cap_1000():
    rng = np.random.default_rng(7)
    xyz = np.concatenate([rng.normal(0, 0.002, (2600, 3)), rng.random((400, 3)) + 1]).astype(np.float32)
    rsl = _ballquery_case(xyz, np.zeros(len(xyz), np.int32), 0.04)
    assert (rsl[:2600, 1] == 1000).all()


def test_octree_ball_query(golden):
    rng = np.random.default_rng(11)
    for xyz, r in ((_blob_cloud(rng, 10, 400, 3000, sigma=0.3) * np.float32(40), 0.9 * 3),
                   (rng.normal(0, 0.01, (1800, 3)).astype(np.float32), 1.0)):
        xyz = np.ascontiguousarray(xyz, np.float32)
        xyzwhl = np.concatenate([(xyz.max(0) + xyz.min(0)) / 2, xyz.max(0) - xyz.min(0)]).astype(np.float32)
        boxes, pt_inds, pt_sl = oracle.build_and_export_octree(xyz, xyzwhl, 3)   # pinned vs _ref CPU build
        ridx, rsl = ref_gpu.octree_ball_query(t(xyz), t(boxes), t(pt_inds), t(pt_sl), 3, r)
        ridx, rsl = ridx.cpu().numpy(), rsl.cpu().numpy()
        oidx, osl = oracle.octree_ball_query(xyz, 3, r)
        assert _same_lists(oidx, osl, ridx, rsl)                                          # (i)
        idx, sl = ops.octree_ball_query(t(xyz), 3, r)
        assert _same_lists(idx.cpu().numpy(), sl.cpu().numpy(), ridx, rsl)                # (ii)


# ----------------------------------------------------------------------------- segment ops / ROI pool
@pytest.mark.parametrize('C', [3, 32, 64, 300])
def test_sec_ops_and_avg_pool(C):
    rng = np.random.default_rng(200 + C)
    off = _segments(rng, 57, 1, 900)
    x = rng.standard_normal((off[-1], C)).astype(np.float32)
    for name in ('sec_min', 'sec_max', 'sec_mean', 'global_avg_pool'):
        ref = getattr(ref_gpu, name)(t(x), t(off)).cpu().numpy()
        assert np.array_equal(getattr(oracle, name)(x, off), ref), name                   # (i)
        assert np.array_equal(getattr(ops, name)(t(x), t(off)).cpu().numpy(), ref), name  # (ii)
    g = rng.standard_normal((len(off) - 1, C)).astype(np.float32)
    ref_bp = ref_gpu.global_avg_pool_bp(t(g), t(off), x.shape[0]).cpu().numpy()
    assert np.array_equal(oracle.global_avg_pool_bp(g, off, x.shape[0]), ref_bp)
    xt = t(x).requires_grad_(True)
    ops.global_avg_pool(xt, t(off)).backward(t(g))
    assert np.array_equal(xt.grad.cpu().numpy(), ref_bp)


def test_sec_ops_on_proposal_coords():
    """the shape the model uses them at (clusters_voxelization, softgroup.py:679-680): [S,3] coords"""
    rng = np.random.default_rng(31)
    off = _segments(rng, 400, 50, 3000)
    x = (rng.random((off[-1], 3)) * 8).astype(np.float32)
    for name in ('sec_min', 'sec_max'):
        ref = getattr(ref_gpu, name)(t(x), t(off)).cpu().numpy()
        assert np.array_equal(getattr(oracle, name)(x, off), ref)
        assert np.array_equal(getattr(ops, name)(t(x), t(off)).cpu().numpy(), ref)


# ----------------------------------------------------------------------------- mask IoU / label
def test_mask_iou_and_label():
    rng = np.random.default_rng(23)
    N, nI, nP = 60000, 37, 90
    inst = rng.integers(-1, nI, N).astype(np.int64)
    inst[inst < 0] = -100
    pointnum = np.array([(inst == g).sum() for g in range(nI)], np.int32)
    off = _segments(rng, nP, 20, 1500)
    pidx = rng.integers(0, N, off[-1]).astype(np.int32)
    cls = rng.integers(0, 18, nI).astype(np.int64)
    cls[[3, 9]] = -100
    sig = rng.random(off[-1]).astype(np.float32)
    sig[::97] = 0.5                                   # exactly on the '> 0.5' threshold
    a = (t(pidx), t(off), t(inst), t(pointnum))
    ref = ref_gpu.get_mask_iou_on_cluster(*a)
    assert np.array_equal(oracle.get_mask_iou_on_cluster(pidx, off, inst, pointnum), ref.cpu().numpy())
    assert np.array_equal(ops.get_mask_iou_on_cluster(*a).cpu().numpy(), ref.cpu().numpy())
    ref2 = ref_gpu.get_mask_iou_on_pred(*a, t(sig))
    assert np.array_equal(oracle.get_mask_iou_on_pred(pidx, off, inst, pointnum, sig), ref2.cpu().numpy())
    assert np.array_equal(ops.get_mask_iou_on_pred(*a, t(sig)).cpu().numpy(), ref2.cpu().numpy())
    for thr in (0.0, 0.01, 0.5):
        rml = ref_gpu.get_mask_label(t(pidx), t(off), t(inst), t(cls), t(pointnum), ref, thr).cpu().numpy()
        assert np.array_equal(oracle.get_mask_label(pidx, off, inst, cls, pointnum, ref.cpu().numpy(), thr), rml)
        ml = ops.get_mask_label(t(pidx), t(off), t(inst), t(cls), t(pointnum), ref, thr)
        assert np.array_equal(ml.cpu().numpy(), rml)
