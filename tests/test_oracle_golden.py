"""The CPU oracle (oracle/sg_oracle*.c) against the golden vectors generated from the
reference itself (tests/golden/make_golden.py) -- the 'pin' of the oracle."""
import numpy as np

import oracle


def test_voxelize_idx_matches_reference(golden):
    g = golden('voxelize_idx')
    for i in range(int(g['vox_ncases'])):
        for mode in (4, 3):
            p = f'vox{i}_m{mode}_'
            oc, im, om = oracle.voxelization_idx(g[p + 'coords'], int(g[p + 'batch']), mode)
            assert np.array_equal(oc, g[p + 'out_coords'])
            assert np.array_equal(im, g[p + 'input_map'])
            assert np.array_equal(om, g[p + 'output_map'])


def test_bfs_cluster_matches_reference(golden):
    g = golden('bfs_cluster')
    for k in range(int(g['bfs_ncases'])):
        p = f'bfs{k}_'
        ci, co = oracle.bfs_cluster(g[p + 'mean'], g[p + 'idx'], g[p + 'start_len'],
                                    float(g[p + 'thr']), int(g[p + 'cid']))
        assert np.array_equal(ci, g[p + 'cluster_idxs'])
        assert np.array_equal(co, g[p + 'cluster_offsets'])


def test_octree_build_matches_reference(golden):
    g = golden('octree')
    for k in range(int(g['oct_ncases'])):
        p = f'oct{k}_'
        b, pi, ps = oracle.build_and_export_octree(g[p + 'points'], g[p + 'xyzwhl'], 3)
        assert np.array_equal(b, g[p + 'boxes'])
        assert np.array_equal(pi, g[p + 'pt_inds'])
        assert np.array_equal(ps, g[p + 'pt_start_len'])


def test_sparse_conv_matches_dense_torch(golden):
    """fp32, tolerance 1e-4 (summation order differs from the dense conv by design)."""
    g = golden('sparse_conv_dense')
    idx, f, shape = g['indices'], g['feats'], g['shape']
    nbr = oracle.subm_rulebook(idx, shape)
    np.testing.assert_allclose(oracle.subm_conv3d(f, nbr, g['W_subm']), g['subm_out'], atol=1e-4,
                               rtol=1e-4)
    oi, in2out, child, oshape = oracle.down_rulebook(idx, shape)
    assert oshape == [4, 4, 4]
    assert ((in2out < 0) == (idx[:, 1:] >= 8).any(1)).all()      # odd-extent last plane dropped
    down = oracle.sparse_conv3d_k2s2(f, child, g['W_down'])
    dd = g['down_dense']
    np.testing.assert_allclose(down, dd[oi[:, 0], oi[:, 1], oi[:, 2], oi[:, 3]], atol=1e-4,
                               rtol=1e-4)
    # every non-zero dense output site is an active output row
    mask = np.zeros(dd.shape[:4], bool)
    mask[oi[:, 0], oi[:, 1], oi[:, 2], oi[:, 3]] = True
    assert np.abs(dd[~mask]).max() == 0
    # inverse conv consumes the *dense* down output sampled at the active rows
    inv = oracle.inverse_conv3d_k2(dd[oi[:, 0], oi[:, 1], oi[:, 2], oi[:, 3]], idx, in2out,
                                   g['W_inv'])
    np.testing.assert_allclose(inv, g['inverse_out'], atol=1e-4, rtol=1e-4)


def test_ballquery_properties():
    """No reference CPU build exists for ballquery_batch_p: check the restatement's
    invariants (ascending lists, self included, symmetric, batch separation) and that the
    octree variant returns the same neighbour sets."""
    rng = np.random.default_rng(3)
    n = 1500
    xyz = rng.random((n, 3)).astype(np.float32)
    bi = np.sort(rng.integers(0, 3, n)).astype(np.int32)
    bo = np.concatenate([[0], np.cumsum(np.bincount(bi, minlength=3))]).astype(np.int32)
    idx, sl = oracle.ballquery_batch_p(xyz, bi, bo, 0.1, 2)   # mean_active 2 forces the retry loop
    assert sl[:, 1].sum() == idx.shape[0]
    nb = [idx[s:s + l] for s, l in sl]
    for p in range(n):
        assert (np.diff(nb[p]) > 0).all() and p in nb[p]
        assert (bi[nb[p]] == bi[p]).all()
        d2 = ((xyz[nb[p]] - xyz[p])**2).sum(1)
        assert (d2 < 0.1 * 0.1 + 1e-6).all()
    for p in range(0, n, 7):
        for q in nb[p]:
            assert p in nb[q]
    i1, s1 = oracle.octree_ball_query(xyz, 5, 0.1)
    i2, s2 = oracle.ballquery_batch_p(xyz, np.zeros(n, np.int32), np.array([0, n], np.int32), 0.1, 5)
    for p in range(n):
        assert np.array_equal(np.sort(i1[s1[p, 0]:s1[p, 0] + s1[p, 1]]), i2[s2[p, 0]:s2[p, 0] + s2[p, 1]])
