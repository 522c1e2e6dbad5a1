"""GPU parity tests of the grouping-head operators: HIP path (through the C ABI) vs the CPU
oracle on the same seeded inputs, plus the golden vectors generated from the reference.

Bar: bit-exact for every integer/index product and for the order-preserving fp32 sums;
exact for min/max and IoU (integer counts, f64 quotient)."""
import numpy as np
import pytest
import torch

import oracle
from softgroup_amd import ops

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def t(a, dtype=None):
    x = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return x if dtype is None else x.to(dtype)


# ----------------------------------------------------------------------------- voxelisation
def test_voxelization_idx_gpu_golden(golden):
    g = golden('voxelize_idx')
    for i in range(int(g['vox_ncases'])):
        for mode in (4, 3):
            p = f'vox{i}_m{mode}_'
            oc, im, om = ops.voxelization_idx(t(g[p + 'coords']), int(g[p + 'batch']), mode)
            assert oc.is_cuda and oc.dtype == torch.int64 and om.dtype == torch.int32
            assert np.array_equal(oc.cpu().numpy(), g[p + 'out_coords'])
            assert np.array_equal(im.cpu().numpy(), g[p + 'input_map'])
            assert np.array_equal(om.cpu().numpy(), g[p + 'output_map'])


@pytest.mark.parametrize('n,hi,B', [(150000, 300, 1), (100000, 20, 200), (5000, 2, 1), (1, 5, 1)])
def test_voxelization_idx_gpu_vs_oracle(n, hi, B):
    rng = np.random.default_rng(n + hi)
    c = rng.integers(0, hi, (n, 4)).astype(np.int64)
    c[:, 0] = np.sort(rng.integers(0, B, n))
    oc, im, om = ops.voxelization_idx(t(c), B)
    roc, rim, rom = oracle.voxelization_idx(c, B)
    assert np.array_equal(oc.cpu().numpy(), roc)
    assert np.array_equal(im.cpu().numpy(), rim)
    assert np.array_equal(om.cpu().numpy(), rom)


def test_voxelization_idx_gpu_modes_and_empty():
    c = torch.tensor([[0, 1, 1, 1], [0, 2, 2, 2], [0, 1, 1, 1]], device=DEV)
    assert ops.voxelization_idx(c, 1, 1)[2].tolist() == [[1, 0], [1, 1]]
    assert ops.voxelization_idx(c, 1, 2)[2].tolist() == [[1, 2], [1, 1]]
    oc, im, om = ops.voxelization_idx(torch.zeros((0, 4), dtype=torch.int64, device=DEV), 1)
    assert oc.shape == (0, 4) and im.shape == (0,) and om.shape[0] == 0


@pytest.mark.parametrize('C', [6, 32, 3, 1, 33])
def test_voxelization_fp_bp_bit_exact(C):
    rng = np.random.default_rng(C)
    n = 40000
    c = rng.integers(0, 24, (n, 4)).astype(np.int64)
    c[:, 0] = 0
    _, _, om = oracle.voxelization_idx(c, 1)
    feats = rng.standard_normal((n, C)).astype(np.float32)
    for mode in (4, 3):
        ref = oracle.voxelization(feats, om, mode)
        f = t(feats).requires_grad_(True)
        out = ops.voxelization(f, t(om), mode)
        assert np.array_equal(out.detach().cpu().numpy(), ref)          # bit exact
        g = rng.standard_normal(ref.shape).astype(np.float32)
        out.backward(t(g))
        assert np.array_equal(f.grad.cpu().numpy(), oracle.voxelization_bp(g, om, n, mode))


# ----------------------------------------------------------------------------- ball query
def _blob_cloud(rng, n_blobs, per_blob, n_noise, sigma=0.03):
    ctr = rng.random((n_blobs, 3)) * np.array([6, 5, 2.7])
    pts = [ctr[i] + rng.normal(0, sigma, (per_blob, 3)) for i in range(n_blobs)]
    pts.append(rng.random((n_noise, 3)) * np.array([6, 5, 2.7]))
    xyz = np.concatenate(pts).astype(np.float32)
    return xyz[rng.permutation(len(xyz))]


def _check_ballquery(xyz, bi, radius):
    n = xyz.shape[0]
    B = int(bi.max()) + 1 if n else 1
    bo = np.concatenate([[0], np.cumsum(np.bincount(bi, minlength=B))]).astype(np.int32)
    idx, sl = ops.ballquery_batch_p(t(xyz), t(bi), t(bo), radius, 300)
    ridx, rsl = oracle.ballquery_batch_p(xyz, bi, bo, radius, 300)
    idx, sl = idx.cpu().numpy(), sl.cpu().numpy()
    assert np.array_equal(sl[:, 1], rsl[:, 1])
    assert np.array_equal(sl[:, 0], np.concatenate([[0], np.cumsum(sl[:-1, 1])]))   # ascending CSR
    assert np.array_equal(idx, ridx)     # oracle also lays lists out in point order
    return idx, sl


def test_ballquery_vs_oracle_blobs():
    rng = np.random.default_rng(2)
    xyz = _blob_cloud(rng, 12, 300, 2000)
    n = len(xyz)
    bi = np.sort(rng.integers(0, 2, n)).astype(np.int32)
    _check_ballquery(xyz, bi, 0.04)


def test_ballquery_negative_coords_and_cell_boundaries():
    rng = np.random.default_rng(5)
    # lattice points exactly on multiples of the radius: stresses strict '<' and cell edges
    g = np.stack(np.meshgrid(*[np.arange(-6, 6)] * 3, indexing='ij'), -1).reshape(-1, 3)
    xyz = (g * 0.05).astype(np.float32)
    xyz = np.concatenate([xyz, xyz + np.float32(1e-4), (rng.random((500, 3)) - 0.5).astype(np.float32)])
    _check_ballquery(xyz, np.zeros(len(xyz), np.int32), 0.05)


def test_ballquery_cap_1000_keeps_smallest_indices():
    rng = np.random.default_rng(7)
    # 2600 points inside a 1 cm ball: every list hits the cap; > LDS stage (2048) -> slow path
    xyz = (rng.normal(0, 0.002, (2600, 3))).astype(np.float32)
    xyz = np.concatenate([xyz, rng.random((400, 3)).astype(np.float32) + 1]).astype(np.float32)
    idx, sl = _check_ballquery(xyz, np.zeros(len(xyz), np.int32), 0.04)
    assert (sl[:2600, 1] == 1000).all()
    # 1500 points: cap hit, but fits the LDS stage (fast path, rank < 1000 filter)
    xyz2 = (rng.normal(0, 0.002, (1500, 3))).astype(np.float32)
    _check_ballquery(xyz2, np.zeros(1500, np.int32), 0.04)


def test_ballquery_empty_and_single():
    z = torch.zeros((0, 3), device=DEV)
    idx, sl = ops.ballquery_batch_p(z, torch.zeros(0, dtype=torch.int32, device=DEV),
                                    torch.zeros(2, dtype=torch.int32, device=DEV), 0.04, 300)
    assert idx.numel() == 0 and sl.shape == (0, 2)
    idx, sl = ops.ballquery_batch_p(torch.ones((1, 3), device=DEV),
                                    torch.zeros(1, dtype=torch.int32, device=DEV),
                                    torch.tensor([0, 1], dtype=torch.int32, device=DEV), 0.04, 300)
    assert idx.tolist() == [0] and sl.tolist() == [[0, 1]]


def test_octree_ball_query_vs_oracle():
    rng = np.random.default_rng(11)
    xyz = _blob_cloud(rng, 10, 400, 3000, sigma=0.3) * np.float32(40)
    idx, sl = ops.octree_ball_query(t(xyz), 3, 0.9 * 3)
    ridx, rsl = oracle.octree_ball_query(xyz, 3, 0.9 * 3)
    assert np.array_equal(sl.cpu().numpy(), rsl)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    # cap: dense clump -> lists stop at the first 1000 in leaf order
    clump = rng.normal(0, 0.01, (1800, 3)).astype(np.float32)
    idx, sl = ops.octree_ball_query(t(clump), 3, 1.0)
    ridx, rsl = oracle.octree_ball_query(clump, 3, 1.0)
    assert (rsl[:, 1] == 1000).all()
    assert np.array_equal(sl.cpu().numpy(), rsl) and np.array_equal(idx.cpu().numpy(), ridx)


# ----------------------------------------------------------------------------- clustering
def test_bfs_cluster_golden(golden):
    g = golden('bfs_cluster')
    for k in range(int(g['bfs_ncases'])):
        p = f'bfs{k}_'
        for dev in ('cpu', DEV):        # CPU tensors in -> CPU out (reference contract); CUDA -> CUDA
            ci, co = ops.bfs_cluster(torch.from_numpy(g[p + 'mean']),
                                     torch.from_numpy(g[p + 'idx']).to(dev),
                                     torch.from_numpy(g[p + 'start_len']).to(dev),
                                     float(g[p + 'thr']), int(g[p + 'cid']))
            assert ci.device.type == torch.device(dev).type and ci.dtype == torch.int32
            assert np.array_equal(ci.cpu().numpy().reshape(-1, 2), g[p + 'cluster_idxs']), k
            assert np.array_equal(co.cpu().numpy(), g[p + 'cluster_offsets']), k


def test_bfs_cluster_vs_oracle_large_and_order():
    rng = np.random.default_rng(13)
    xyz = _blob_cloud(rng, 40, 1000, 10000)
    n = len(xyz)
    bi = np.zeros(n, np.int32)
    bo = np.array([0, n], np.int32)
    idx, sl = ops.ballquery_batch_p(t(xyz), t(bi), t(bo), 0.04, 300)
    mean = torch.tensor([-1.0, 2000.0])
    for cid, thr in ((0, 100.0), (1, 0.05)):
        ci, co = ops.bfs_cluster(mean, idx, sl, thr, cid)
        rci, rco = oracle.bfs_cluster(mean.numpy(), idx.cpu().numpy(), sl.cpu().numpy(), thr, cid)
        assert np.array_equal(co.cpu().numpy(), rco)
        assert np.array_equal(ci.cpu().numpy(), rci)       # membership AND member order
    assert len(rco) - 1 >= 40


def test_bfs_cluster_thin_levels_on_one_wave_equal_the_workgroup_path(monkeypatch):
    """bfs_emit_kernel replays runs of thin levels (<= 64 frontier nodes, <= 512 edges) on ONE wave: chains
    (hundreds of levels of 1-3 nodes), small blobs, levels that alternate between thin and fat, a blob whose
    middle levels exceed 512 edges -- membership and BFS order against the oracle, with the single-wave runs
    (SG_BFS_THIN=1; measured slower, off by default) and without."""
    rng = np.random.default_rng(31)
    parts = []
    # chains: points 1.5 cm apart along a wiggly line -> 2-4 neighbours each, ~400 levels
    for k in range(3):
        tt = np.arange(400 + 100 * k) * 0.015
        parts.append(np.stack([tt, 0.02 * np.sin(7 * tt) + 2.0 * k, np.zeros_like(tt)], 1))
    # small blobs (thin throughout), one dense blob (fat middle levels), dumbbells (thin - fat - thin)
    for k in range(12):
        parts.append(rng.normal(0, 0.02, (150 + 20 * k, 3)) + np.array([10.0 + k, 0, 0]))
    parts.append(rng.normal(0, 0.03, (6000, 3)) + np.array([30.0, 0, 0]))
    for k in range(3):
        bar = np.stack([np.linspace(0, 0.6, 60), np.zeros(60), np.zeros(60)], 1)
        ends = [rng.normal(0, 0.015, (800, 3)), rng.normal(0, 0.015, (800, 3)) + np.array([0.6, 0, 0])]
        parts.append(np.concatenate([bar] + ends) + np.array([40.0 + 3 * k, 0, 0]))
    xyz = np.concatenate(parts).astype(np.float32)
    xyz = xyz[rng.permutation(len(xyz))]
    n = len(xyz)
    idx, sl = ops.ballquery_batch_p(t(xyz), t(np.zeros(n, np.int32)), t(np.array([0, n], np.int32)), 0.04, 300)
    mean = torch.tensor([-1.0])
    rci, rco = oracle.bfs_cluster(mean.numpy(), idx.cpu().numpy(), sl.cpu().numpy(), 20.0, 0)
    assert len(rco) - 1 >= 15
    for thin in ("2", "1", "0"):
        monkeypatch.setenv('SG_BFS_THIN', thin)
        ci, co = ops.bfs_cluster(mean, idx, sl, 20.0, 0)
        assert np.array_equal(co.cpu().numpy(), rco), thin
        assert np.array_equal(ci.cpu().numpy(), rci), thin       # membership AND member order


def test_bfs_cluster_directed_lists_capped():
    """cap-hit regime: lists keep the 1000 smallest indices -> asymmetric graph -> the reference
    semantics is directed reachability from ascending seeds."""
    rng = np.random.default_rng(17)
    a = rng.normal(0, 0.004, (1800, 3))
    b = rng.normal(0, 0.004, (1500, 3)) + np.array([0.03, 0, 0])
    c = rng.random((700, 3)) + 1.0
    xyz = np.concatenate([a, b, c]).astype(np.float32)
    xyz = xyz[rng.permutation(len(xyz))]
    n = len(xyz)
    idx, sl = ops.ballquery_batch_p(t(xyz), t(np.zeros(n, np.int32)), t(np.array([0, n], np.int32)),
                                    0.04, 300)
    assert (sl[:, 1] == 1000).any()
    mean = torch.tensor([-1.0])
    ci, co = ops.bfs_cluster(mean, idx, sl, 2.0, 0)
    rci, rco = oracle.bfs_cluster(mean.numpy(), idx.cpu().numpy(), sl.cpu().numpy(), 2.0, 0)
    assert np.array_equal(co.cpu().numpy(), rco) and np.array_equal(ci.cpu().numpy(), rci)
    # unsorted lists (octree order) take the linear membership test
    oidx, osl = ops.octree_ball_query(t(xyz), 3, 0.04)
    ci, co = ops.bfs_cluster(mean, oidx, osl, 2.0, 0)
    rci, rco = oracle.bfs_cluster(mean.numpy(), oidx.cpu().numpy(), osl.cpu().numpy(), 2.0, 0)
    assert np.array_equal(co.cpu().numpy(), rco) and np.array_equal(ci.cpu().numpy(), rci)


def test_bfs_cluster_bigger_than_the_lds_claim_array():
    """one 22 500-point cluster (> 16 384: claims live in global memory, every level takes the
    generic path) next to small ones that take the fast path -- membership and BFS order exact."""
    g = np.stack(np.meshgrid(np.arange(150), np.arange(150), indexing='ij'), -1).reshape(-1, 2)
    sheet = np.concatenate([g * 0.02, np.zeros((len(g), 1))], 1)
    rng = np.random.default_rng(23)
    small = [rng.normal(0, 0.01, (300, 3)) + np.array([5.0 + 0.5 * i, 1.0, 1.0]) for i in range(4)]
    xyz = np.concatenate([sheet] + small).astype(np.float32)
    xyz = xyz[rng.permutation(len(xyz))]
    n = len(xyz)
    idx, sl = ops.ballquery_batch_p(t(xyz), t(np.zeros(n, np.int32)), t(np.array([0, n], np.int32)),
                                    0.03, 300)
    mean = torch.tensor([-1.0])
    ci, co = ops.bfs_cluster(mean, idx, sl, 50.0, 0)
    rci, rco = oracle.bfs_cluster(mean.numpy(), idx.cpu().numpy(), sl.cpu().numpy(), 50.0, 0)
    assert np.diff(rco).max() == 22500 and len(rco) - 1 == 5
    assert np.array_equal(co.cpu().numpy(), rco) and np.array_equal(ci.cpu().numpy(), rci)
    # the multi-workgroup replay's grid barrier is bounded; if it ever gives up (workgroups not
    # co-resident) the giant clusters are replayed by the per-cluster kernel, gated on the device.
    # SG_BFS_FORCE_FALLBACK sets the failure word up front: the result must be the same.
    import os
    os.environ['SG_BFS_FORCE_FALLBACK'] = '1'
    try:
        ci2, co2 = ops.bfs_cluster(mean, idx, sl, 50.0, 0)
    finally:
        del os.environ['SG_BFS_FORCE_FALLBACK']
    assert np.array_equal(co2.cpu().numpy(), rco) and np.array_equal(ci2.cpu().numpy(), rci)
    # the default replay is the LOCAL form (a workgroup keeps its own children, one rendezvous per level) in
    # front of the round-5 form in front of the per-cluster kernel: without the LOCAL form, with the LOCAL
    # form's fail word set up front (bit 1: the round-5 form redoes the clusters), and with both set
    for local, fallback in (('0', None), ('1', '2'), ('1', '3')):
        os.environ['SG_BFS_BIG_LOCAL'] = local
        if fallback:
            os.environ['SG_BFS_FORCE_FALLBACK'] = fallback
        try:
            ci3, co3 = ops.bfs_cluster(mean, idx, sl, 50.0, 0)
        finally:
            del os.environ['SG_BFS_BIG_LOCAL']
            os.environ.pop('SG_BFS_FORCE_FALLBACK', None)
        assert np.array_equal(co3.cpu().numpy(), rco), (local, fallback)
        assert np.array_equal(ci3.cpu().numpy(), rci), (local, fallback)


def test_bfs_cluster_giant_with_fat_levels_both_replay_forms():
    """one 40 000-point slab whose BFS levels have 10^5..10^6 edges (a workgroup's share exceeds the
    cached level: the multi-workgroup replay takes its chunked path, frontier regions come out of the
    shared pool) and the thin sheet above (always the cached level, private slices): membership and BFS
    order exact, in the default form of the replay (LOCAL), the round-5 form and the round-4 form."""
    import os
    rng = np.random.default_rng(29)
    slab = rng.random((40000, 3)) * np.array([1.0, 1.0, 0.02])
    g = np.stack(np.meshgrid(np.arange(150), np.arange(150), indexing='ij'), -1).reshape(-1, 2)
    sheet = np.concatenate([g * 0.02 + 3.0, np.zeros((len(g), 1))], 1)
    xyz = np.concatenate([slab, sheet]).astype(np.float32)
    xyz = xyz[rng.permutation(len(xyz))]
    n = len(xyz)
    idx, sl = ops.ballquery_batch_p(t(xyz), t(np.zeros(n, np.int32)), t(np.array([0, n], np.int32)),
                                    0.04, 300)
    assert float(sl[:, 1].float().mean()) > 60
    mean = torch.tensor([-1.0])
    rci, rco = oracle.bfs_cluster(mean.numpy(), idx.cpu().numpy(), sl.cpu().numpy(), 50.0, 0)
    assert sorted(np.diff(rco).tolist()) == [22500, 40000]
    for form in ({}, {'SG_BFS_BIG_LOCAL': '0'}, {'SG_BFS_BIG_FAST': '0'}, {'SG_BFS_BIG_LOCAL_WGS': '8'},
                 {'SG_BFS_BIG_LOCAL_WGS': '64', 'SG_BFS_BIG_LOCAL_EVERY': '16'}, {'SG_BFS_BIG_LOCAL_NOVIS': '1'},
                 {'SG_BFS_BIG_LOCAL_EVERY': '1', 'SG_BFS_BIG_LOCAL_WGS': '24'}):
        # default: the LOCAL form (the slab's fat levels work out the edges beyond its level cache twice; where
        # a workgroup's share of a frontier exceeds what it holds it gives up and the round-5 form redoes both
        # clusters -- 8 workgroups); without it; the round-4 form; few / many workgroups, rare re-partitions;
        # without the LDS visited filter (the path of scenes above 524 288 points); re-partition at every level
        os.environ.update(form)
        try:
            ci, co = ops.bfs_cluster(mean, idx, sl, 50.0, 0)
        finally:
            for k in form:
                del os.environ[k]
        assert np.array_equal(co.cpu().numpy(), rco), form
        assert np.array_equal(ci.cpu().numpy(), rci), form


def test_bfs_cluster_empty_and_all_dropped():
    mean = torch.tensor([-1.0])
    ci, co = ops.bfs_cluster(mean, torch.zeros(0, dtype=torch.int32, device=DEV),
                             torch.zeros((0, 2), dtype=torch.int32, device=DEV), 1.0, 0)
    assert ci.shape == (0, 2) and co.tolist() == [0]
    idx = torch.arange(5, dtype=torch.int32, device=DEV)
    sl = torch.stack([torch.arange(5), torch.ones(5, dtype=torch.long)], 1).int().to(DEV)
    ci, co = ops.bfs_cluster(mean, idx, sl, 2.0, 0)        # singletons < thr
    assert ci.shape == (0, 2) and co.tolist() == [0]
    ci, co = ops.bfs_cluster(mean, idx, sl, 1.0, 0)
    assert ci.tolist() == [[i, i] for i in range(5)] and co.tolist() == list(range(6))


def test_bfs_cluster_multi_segment_equals_per_class():
    rng = np.random.default_rng(19)
    parts, segs = [], []
    for s in range(3):
        x = _blob_cloud(rng, 6, 250, 800)
        parts.append(x)
        segs.append(np.full(len(x), s, np.int32))
    xyz = np.concatenate(parts)
    seg = np.concatenate(segs)
    n = len(xyz)
    # segments are kept apart by giving each its own "batch" id
    idx, sl = ops.ballquery_batch_p(t(xyz), t(seg), t(np.array([0, n], np.int32)), 0.04, 300)
    thr = np.array([50.0, 120.0, 10.0], np.float32)
    ci, co = ops.bfs_cluster_segments(idx, sl, t(thr), t(seg))
    # reference: one call per class on the class-local sub-problem, then merged (softgroup.py:464-473)
    exp_idx, exp_off, base, nclu = [], [0], 0, 0
    for s in range(3):
        m = len(parts[s])
        li, lsl = oracle.ballquery_batch_p(parts[s], np.zeros(m, np.int32), np.array([0, m], np.int32),
                                           0.04, 300)
        rci, rco = oracle.bfs_cluster(np.array([-1.0], np.float32), li, lsl, float(thr[s]), 0)
        rci = rci.copy()
        rci[:, 0] += nclu
        rci[:, 1] += base
        exp_idx.append(rci)
        last = exp_off[-1]
        exp_off.extend((rco[1:] + last).tolist())
        base += m
        nclu += len(rco) - 1
    assert np.array_equal(ci.cpu().numpy(), np.concatenate(exp_idx))
    assert np.array_equal(co.cpu().numpy(), np.array(exp_off, np.int32))


# ----------------------------------------------------------------------------- segment ops
def _segments(rng, nP, lo, hi):
    lens = rng.integers(lo, hi, nP)
    return np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)


@pytest.mark.parametrize('C', [3, 32, 64, 300])
def test_sec_min_max_mean_avgpool(C):
    rng = np.random.default_rng(C)
    off = _segments(rng, 57, 1, 900)
    off[5] = off[4]                                     # an empty segment
    x = rng.standard_normal((off[-1], C)).astype(np.float32)
    assert np.array_equal(ops.sec_min(t(x), t(off)).cpu().numpy(), oracle.sec_min(x, off))
    assert np.array_equal(ops.sec_max(t(x), t(off)).cpu().numpy(), oracle.sec_max(x, off))
    np.testing.assert_array_equal(ops.sec_mean(t(x), t(off)).cpu().numpy(), oracle.sec_mean(x, off))
    xt = t(x).requires_grad_(True)
    out = ops.global_avg_pool(xt, t(off))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), oracle.global_avg_pool(x, off))
    g = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(t(g))
    ref = oracle.global_avg_pool_bp(g, off, x.shape[0])
    got = xt.grad.cpu().numpy()
    ok = np.isfinite(ref).all(1)                        # rows of the empty segment: none exist
    np.testing.assert_array_equal(got[ok], ref[ok])


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
    iou = ops.get_mask_iou_on_cluster(t(pidx), t(off), t(inst), t(pointnum))
    ref = oracle.get_mask_iou_on_cluster(pidx, off, inst, pointnum)
    assert np.array_equal(iou.cpu().numpy(), ref)
    iou2 = ops.get_mask_iou_on_pred(t(pidx), t(off), t(inst), t(pointnum), t(sig))
    assert np.array_equal(iou2.cpu().numpy(), oracle.get_mask_iou_on_pred(pidx, off, inst, pointnum, sig))
    for thr in (0.0, 0.01, 0.5):
        ml = ops.get_mask_label(t(pidx), t(off), t(inst), t(cls), t(pointnum), iou, thr)
        assert np.array_equal(ml.cpu().numpy(),
                              oracle.get_mask_label(pidx, off, inst, cls, pointnum, ref, thr))


# ----------------------------------------------------------------------------- instance runs
def test_instance_npoint_and_runs_vs_dense_masks():
    """csrc/instances.hip against the reference's dense formulation (softgroup.py:566-590): per
    class a [nProposal, N] 0/1 matrix, row sums, and the runs of each kept row (what rle_encode
    turns into text).  Long runs that cross 32-bit word boundaries, empty and full rows."""
    from softgroup_amd import _lib as L
    lib = L.lib()
    rng = np.random.default_rng(41)
    N, nP, nc, stride = 5000, 23, 5, 6
    # proposal p = a few contiguous point ranges (long runs) + scattered points; proposal-major pairs
    pairs = []
    for p in range(nP):
        pts = set()
        for _ in range(rng.integers(1, 4)):
            a = int(rng.integers(0, N - 400))
            pts.update(range(a, a + int(rng.integers(1, 400))))
        pts.update(rng.integers(0, N, 30).tolist())
        if p == 3:
            pts = set(range(N))                        # a full row
        q = rng.permutation(sorted(pts))               # BFS order is not sorted
        pairs.append(np.stack([np.full(len(q), p), q], 1))
    pairs = np.concatenate(pairs).astype(np.int32)
    S = len(pairs)
    ms = rng.standard_normal((S, stride)).astype(np.float32)
    ms[pairs[:, 0] == 5] = -9.0                        # proposal 5: nothing above the threshold
    thr = -0.5
    d_pairs, d_ms = t(pairs), t(ms)                    # named: alive until the launches are enqueued
    npoint = torch.empty((nP, nc), dtype=torch.int32, device=DEV)
    L.check(lib.sg_instance_npoint(L.ptr(d_pairs), L.ptr(d_ms), S, stride, nc, thr, nP, L.ptr(npoint),
                                   L.stream()), 'sg_instance_npoint')
    dense = np.zeros((nc, nP, N), np.int32)
    for i in range(nc):
        on = ms[:, i] > thr
        dense[i, pairs[on, 0], pairs[on, 1]] = 1
    assert np.array_equal(npoint.cpu().numpy(), dense.sum(2).T)
    keep = dense.sum(2) >= 40                          # [nc, nP]
    keep[1, 7] = False
    kept = np.argwhere(keep)                           # class-major
    inst_of = np.full((nc, nP), -1, np.int32)
    inst_of[kept[:, 0], kept[:, 1]] = np.arange(len(kept))
    cap = int(dense.sum(2)[keep].sum())
    starts = torch.empty(cap, dtype=torch.int32, device=DEV)
    ends = torch.empty(cap, dtype=torch.int32, device=DEV)
    bounds = torch.empty(len(kept) + 1, dtype=torch.int64, device=DEV)
    ws = L.workspace(lib.sg_instance_runs_workspace_bytes(len(kept), N), DEV)
    d_inst = t(inst_of)
    L.check(lib.sg_instance_runs(L.ptr(d_pairs), L.ptr(d_ms), S, stride, nc, thr, L.ptr(d_inst), nP,
                                 len(kept), N, L.ptr(starts), L.ptr(ends), L.ptr(bounds), cap,
                                 L.ptr(ws), ws.numel(), L.stream()), 'sg_instance_runs')
    b = bounds.cpu().numpy()
    st, en = starts.cpu().numpy(), ends.cpu().numpy()
    assert b[0] == 0 and b[-1] <= cap
    for k, (i, p) in enumerate(kept):
        row = np.concatenate([[0], dense[i, p], [0]])
        edges = np.flatnonzero(row[1:] != row[:-1])
        assert np.array_equal(st[b[k]:b[k + 1]], edges[0::2]), k
        assert np.array_equal(en[b[k]:b[k + 1]], edges[1::2]), k
    # the text of the same runs, written on the device, equals rle_encode of the dense rows
    from softgroup_amd.util import rle_encode, rle_text_to_dicts
    tcap = int(lib.sg_rle_format_device_text_bytes(cap, N))
    text = torch.empty(tcap, dtype=torch.uint8, device=DEV)
    text_off = torch.empty(len(kept) + 1, dtype=torch.int64, device=DEV)
    ws2 = L.workspace(lib.sg_rle_format_device_workspace_bytes(cap), DEV)
    L.check(lib.sg_rle_format_device(L.ptr(starts), L.ptr(ends), L.ptr(bounds), len(kept), cap, N,
                                     L.ptr(text), tcap, L.ptr(text_off), L.ptr(ws2), ws2.numel(),
                                     L.stream()), 'sg_rle_format_device')
    got = rle_text_to_dicts(N, text, text_off.cpu().tolist())
    for k, (i, p) in enumerate(kept):
        assert got[k] == rle_encode(dense[i, p]), k


def test_rle_text_on_device_digit_boundaries_and_empty_instances():
    """sg_rle_format_device on hand-made runs: starts / lengths at every power of ten, instances
    without runs (first, middle, last), run capacity larger than the run count."""
    from softgroup_amd import _lib as L
    from softgroup_amd.util import rle_encode_runs, rle_text_to_dicts
    lib = L.lib()
    length = 2_000_000_000
    st, ln = [], []
    pos = 0
    for d in range(0, 9):
        for v in (10 ** d - 1, 10 ** d):              # start+1 = 10^d and 10^d + 1 -> digit steps
            if v < pos:
                continue
            run = 10 ** (d % 4) - (1 if d % 2 else 0) or 1
            st.append(v)
            ln.append(run)
            pos = v + run + 1
    st, ln = np.asarray(st, np.int32), np.asarray(ln, np.int32)
    R = len(st)
    bounds = np.asarray([0, 0, 3, 3, 3, R - 1, R, R], np.int64)      # empty: 0, 2, 3, 6
    n = len(bounds) - 1
    cap = R + 100
    d_st = torch.zeros(cap, dtype=torch.int32, device=DEV)
    d_en = torch.zeros(cap, dtype=torch.int32, device=DEV)
    d_st[:R] = t(st)
    d_en[:R] = t(st + ln)
    d_b = t(bounds)
    tcap = int(lib.sg_rle_format_device_text_bytes(cap, length))
    text = torch.empty(tcap, dtype=torch.uint8, device=DEV)
    text_off = torch.empty(n + 1, dtype=torch.int64, device=DEV)
    ws = L.workspace(lib.sg_rle_format_device_workspace_bytes(cap), DEV)
    L.check(lib.sg_rle_format_device(L.ptr(d_st), L.ptr(d_en), L.ptr(d_b), n, cap, length, L.ptr(text),
                                     tcap, L.ptr(text_off), L.ptr(ws), ws.numel(), L.stream()),
            'sg_rle_format_device')
    got = rle_text_to_dicts(length, text, text_off.cpu().tolist())
    for g in range(n):
        lo, hi = bounds[g], bounds[g + 1]
        assert got[g] == rle_encode_runs(length, st[lo:hi], ln[lo:hi]), g
    # no runs at all
    d_b0 = torch.zeros(3, dtype=torch.int64, device=DEV)
    L.check(lib.sg_rle_format_device(L.ptr(d_st), L.ptr(d_en), L.ptr(d_b0), 2, cap, length, L.ptr(text),
                                     tcap, L.ptr(text_off), L.ptr(ws), ws.numel(), L.stream()),
            'sg_rle_format_device')
    assert rle_text_to_dicts(length, text, text_off[:3].cpu().tolist()) == \
        [dict(length=length, counts='')] * 2
    # a text buffer below the bound is refused
    assert lib.sg_rle_format_device(L.ptr(d_st), L.ptr(d_en), L.ptr(d_b), n, cap, length, L.ptr(text),
                                    tcap - 1, L.ptr(text_off), L.ptr(ws), ws.numel(), L.stream()) != 0


@pytest.mark.parametrize('C', [3, 32, 33, 128])
def test_row_gather_and_bn_relu_glue(C):
    """sg_gather_rows_* (devoxelize, softgroup.py:374,677) and sg_bn_relu_f32 (output_layer,
    softgroup.py:65) against torch indexing / the same affine + ReLU: copies are bit-exact, int32
    and int64 indices, repeated and empty index lists, channel counts off the float4 path."""
    from softgroup_amd import _lib as L
    from softgroup_amd.model.softgroup import _take_rows
    lib = L.lib()
    torch.manual_seed(C)
    M, N = 1237, 4001
    feats = torch.randn(M, C, device=DEV)
    for dt in (torch.int32, torch.int64):
        index = torch.randint(0, M, (N, ), device=DEV).to(dt)
        assert torch.equal(_take_rows(feats, index), feats[index.long()])
        assert _take_rows(feats, index[:0]).shape == (0, C)
    scale = torch.rand(C, device=DEV) + 0.5
    shift = torch.randn(C, device=DEV)
    for relu in (1, 0):
        out = torch.empty_like(feats)
        L.check(lib.sg_bn_relu_f32(L.ptr(feats), L.ptr(scale), L.ptr(shift), M, C, relu, L.ptr(out),
                                   L.stream()), 'sg_bn_relu_f32')
        ref = torch.addcmul(shift, feats, scale)          # one fma per element, like the kernel
        ref = ref.clamp_min(0) if relu else ref
        np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), rtol=0, atol=1e-6)


@pytest.mark.parametrize('n', [1, 37, 3000, 150000])
def test_octree_build_on_the_device_equals_the_host_export(n):
    """sg_octree_build (device) == sg_octree_build_host == the reference's build_and_export_octree
    (tests/golden/octree.npz pins the host build): boxes, leaf order, leaf ranges identical"""
    from softgroup_amd import _lib as L
    lib = L.lib()
    rng = np.random.default_rng(n)
    pts = (rng.standard_normal((n, 3)) * np.array([4.0, 2.0, 0.7])).astype(np.float32)
    if n > 100:
        pts[:50] = pts[0]                           # many points in one leaf
    t = torch.from_numpy(pts)
    mx, mn = t.max(0)[0], t.min(0)[0]
    xyzwhl = torch.cat([(mx + mn) / 2, mx - mn]).contiguous()
    boxes_h = torch.zeros((585, 6), dtype=torch.float32)
    inds_h = torch.zeros(n, dtype=torch.int32)
    sl_h = torch.zeros((512, 2), dtype=torch.int32)
    L.check(lib.sg_octree_build_host(L.ptr(t), L.ptr(xyzwhl), n, 3, L.ptr(boxes_h), L.ptr(inds_h), L.ptr(sl_h)),
            'sg_octree_build_host')
    d = t.cuda()
    boxes = torch.empty((585, 6), dtype=torch.float32, device='cuda')
    inds = torch.empty(n, dtype=torch.int32, device='cuda')
    sl = torch.empty((512, 2), dtype=torch.int32, device='cuda')
    ws = torch.empty(lib.sg_octree_build_workspace_bytes(n), dtype=torch.uint8, device='cuda')
    L.check(lib.sg_octree_build(L.ptr(d), n, L.ptr(boxes), L.ptr(inds), L.ptr(sl), L.ptr(ws), ws.numel(),
                                L.stream()), 'sg_octree_build')
    assert torch.equal(boxes.cpu(), boxes_h)
    assert torch.equal(sl.cpu(), sl_h)
    assert torch.equal(inds.cpu(), inds_h)


def test_octree_ball_query_device_and_host_inputs_agree():
    rng = np.random.default_rng(5)
    pts = torch.from_numpy(rng.random((20000, 3)).astype(np.float32) * 3)
    a_idx, a_sl = ops.octree_ball_query(pts.cuda(), 300, 0.05)      # device build
    b_idx, b_sl = ops.octree_ball_query(pts, 300, 0.05)             # host build (CPU input), GPU walk
    assert torch.equal(a_sl, b_sl) and torch.equal(a_idx, b_idx)


def test_pyramid_inverse_map_equals_the_dense_reference_formulation():
    """ops.pyramid_inverse_map == nonzero of the reference's dense [nProposal, n] matrix
    (softgroup/model/softgroup.py:500-507), rows (proposal, point) ascending"""
    rng = np.random.default_rng(2)
    n_vox, n_pts, n_prop = 5000, 40000, 37
    l2p = torch.from_numpy(rng.integers(0, n_vox, n_pts).astype(np.int32)).cuda()
    vox_prop = rng.integers(-1, n_prop, n_vox)                      # -1: voxel in no proposal
    vox_prop[rng.random(n_vox) < 0.3] = -1
    sel = np.nonzero(vox_prop >= 0)[0]
    order = np.lexsort((sel, vox_prop[sel]))
    pidx = torch.from_numpy(np.stack([vox_prop[sel][order], sel[order]], 1).astype(np.int32)).cuda()
    got_idx, got_off = ops.pyramid_inverse_map(pidx, n_prop, l2p, n_vox)
    dense = torch.zeros((n_prop, n_vox), dtype=torch.int32, device='cuda')
    dense[pidx[:, 0].long(), pidx[:, 1].long()] = 1
    exp = dense[:, l2p.long()].nonzero()
    assert torch.equal(got_idx.long(), exp)
    counts = torch.bincount(exp[:, 0], minlength=n_prop)
    assert torch.equal(got_off.long(), torch.cat([counts.new_zeros(1), torch.cumsum(counts, 0)]))


@pytest.mark.parametrize('channels,n_sem,idx_dtype', [(32, 20, torch.int64), (32, 13, torch.int32), (16, 15, torch.int64),
                                                      (16, 20, None)])
def test_pointwise_heads_equal_the_modules(channels, n_sem, idx_dtype):
    """sg_pointwise_heads (devoxelize gather + semantic_linear + offset_linear + arg-max in one kernel)
    against the module path of forward_backbone (softgroup/model/softgroup.py:374-376,320): the fp32
    reference is the same layers evaluated by torch in float64; tolerance 1e-4 of the output scale
    (north-star tolerance for float features), gathered features and arg-max exact."""
    from softgroup_amd import synthetic
    cfg = dict(synthetic.SCANNET_MODEL_CFG, channels=channels, semantic_classes=n_sem, instance_classes=n_sem - 2)
    cfg['grouping_cfg'] = dict(cfg['grouping_cfg'], class_numpoint_mean=[1.0] * n_sem)
    model = synthetic.build_model(cfg=cfg, seed=3)
    g = torch.Generator().manual_seed(5)
    M, N = 7000, 9001
    vox = (torch.randn(M, channels, generator=g) * 2).to(DEV)
    v2p = None if idx_dtype is None else torch.randint(0, M, (N, ), generator=g).to(DEV).to(idx_dtype)
    with torch.no_grad():
        heads = model._fused_heads(vox)
        assert heads is not None
        sem, off, feats, preds = model._run_fused_heads(heads, vox, v2p)
        ref_feats = vox if v2p is None else vox[v2p.long()]
        assert torch.equal(feats, ref_feats)
        d = ref_feats.double()
        ref_sem = model.semantic_linear.double()(d)
        ref_off = model.offset_linear.double()(d)
        model.float()
    assert sem.shape == (ref_feats.shape[0], n_sem) and off.shape == (ref_feats.shape[0], 3)
    for got, ref in ((sem, ref_sem), (off, ref_off)):
        scale = float(ref.abs().max())
        assert float((got.double() - ref).abs().max()) <= 1e-4 * max(scale, 1.0)
    assert torch.equal(preds, sem.max(1)[1])
    # the model takes the fused path in inference and the modules under autograd, same numbers
    with torch.no_grad():
        model.use_fused_heads = False
        m_sem = model.semantic_linear(ref_feats)
        m_off = model.offset_linear(ref_feats)
    assert float((m_sem - sem).abs().max()) <= 1e-4 * max(float(m_sem.abs().max()), 1.0)
    assert float((m_off - off).abs().max()) <= 1e-4 * max(float(m_off.abs().max()), 1.0)
