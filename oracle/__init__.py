"""CPU oracle for the SoftGroup hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this package.  The product path (``softgroup_amd``) never does.

``build()`` compiles ``sg_oracle.c`` / ``sg_oracle_conv.c`` with gcc into
``oracle/_build/`` (git-ignored, travels to the GPU box); ``lib()`` loads them.
Every function takes/returns numpy arrays and mirrors one reference operator
(see the file:line citations in the C sources).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, '_build')
_SO = os.path.join(BUILD, 'libsg_oracle.so')
_SO_CONV = os.path.join(BUILD, 'libsg_oracle_conv.so')
_libs = {}


def _stale(out, srcs):
    return (not os.path.exists(out)) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(out) for s in srcs)


def _cpu_tag():
    """identifies the host CPU's instruction set: the conv oracle is built with -march=native, and
    the built library travels from the build container to the GPU box"""
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('flags'):
                    import hashlib
                    return hashlib.sha1(' '.join(sorted(line.split(':', 1)[1].split())).encode()).hexdigest()
    except OSError:
        pass
    return 'unknown'


def build(force=False):
    os.makedirs(BUILD, exist_ok=True)
    src = os.path.join(HERE, 'sg_oracle.c')
    if os.path.exists(src) and (force or _stale(_SO, [src])):
        subprocess.check_call(['gcc', '-O2', '-ffp-contract=off', '-fPIC', '-shared', '-std=c11',
                               src, '-o', _SO, '-lm'])
    src = os.path.join(HERE, 'sg_oracle_conv.c')
    stamp = _SO_CONV + '.cpu'
    tag = _cpu_tag()
    built_for = open(stamp).read().strip() if os.path.exists(stamp) else None
    if os.path.exists(src) and (force or _stale(_SO_CONV, [src]) or built_for != tag):
        subprocess.check_call(['gcc', '-O3', '-march=native', '-fopenmp', '-fPIC', '-shared',
                               '-std=c11', src, '-o', _SO_CONV, '-lm'])
        with open(stamp, 'w') as f:
            f.write(tag)
    return _SO, _SO_CONV


def lib(which='ops'):
    if which not in _libs:
        build()
        _libs[which] = C.CDLL(_SO if which == 'ops' else _SO_CONV)
    return _libs[which]


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# --------------------------------------------------------------------------- ops
def voxelization_idx(coords, batchsize, mode=4):
    """functions.py:168-197 -> (output_coords i64[M,ncol], input_map i32[N], output_map i32[M,mA+1])"""
    coords = _i64(coords)
    n, ncol = coords.shape
    input_map = np.zeros(n, np.int32)
    M = C.c_int32(0)
    mA = C.c_int32(0)
    rc = lib().orc_voxelize_idx(_p(coords, C.c_int64), n, ncol, mode, _p(input_map, C.c_int32),
                                C.byref(M), C.byref(mA))
    assert rc == 0
    M, mA = M.value, mA.value
    out_coords = np.zeros((M, ncol), np.int64)
    out_map = np.zeros((M, mA + 1), np.int32)
    lib().orc_voxelize_idx_fill(_p(coords, C.c_int64), n, ncol, mode, _p(input_map, C.c_int32), M,
                                mA, _p(out_coords, C.c_int64), _p(out_map, C.c_int32))
    return out_coords, input_map, out_map


def voxelization(feats, map_rule, mode=4):
    feats = _f32(feats)
    map_rule = _i32(map_rule)
    M, mA1 = map_rule.shape
    Cn = feats.shape[1]
    out = np.zeros((M, Cn), np.float32)
    lib().orc_voxelize_fp(_p(feats, C.c_float), _p(out, C.c_float), _p(map_rule, C.c_int32), M,
                          mA1 - 1, Cn, int(mode == 4))
    return out


def voxelization_bp(d_out, map_rule, N, mode=4):
    d_out = _f32(d_out)
    map_rule = _i32(map_rule)
    M, mA1 = map_rule.shape
    Cn = d_out.shape[1]
    d_feats = np.zeros((N, Cn), np.float32)
    lib().orc_voxelize_bp(_p(d_out, C.c_float), _p(d_feats, C.c_float), _p(map_rule, C.c_int32), M,
                          mA1 - 1, Cn, int(mode == 4))
    return d_feats


def ballquery_batch_p(coords, batch_idxs, batch_offsets, radius, mean_active):
    """functions.py:237-275 incl. the grow-and-retry loop -> (idx i32[nActive], start_len i32[n,2])"""
    coords = _f32(coords)
    batch_idxs = _i32(batch_idxs)
    batch_offsets = _i32(batch_offsets)
    n = coords.shape[0]
    f = lib().orc_ballquery_batch_p
    f.restype = C.c_int64
    while True:
        idx = np.zeros(n * mean_active, np.int32)
        start_len = np.zeros((n, 2), np.int32)
        n_active = f(_p(coords, C.c_float), _p(batch_idxs, C.c_int32), _p(batch_offsets, C.c_int32),
                     n, int(mean_active), C.c_float(radius), _p(idx, C.c_int32),
                     _p(start_len, C.c_int32))
        if n_active <= n * mean_active:
            break
        mean_active = int(n_active // n + 1)
    return idx[:n_active], start_len


def bfs_cluster(class_numpoint_mean, ball_query_idxs, start_len, threshold, class_id):
    cm = _f32(class_numpoint_mean)
    idxs = _i32(ball_query_idxs)
    sl = _i32(start_len)
    N = sl.shape[0]
    ci = np.zeros((max(N, 1), 2), np.int32)
    co = np.zeros(N + 1, np.int32)
    nc = C.c_int32(0)
    sp = C.c_int32(0)
    lib().orc_bfs_cluster(_p(cm, C.c_float), _p(idxs, C.c_int32), _p(sl, C.c_int32), N,
                          C.c_float(threshold), int(class_id), _p(ci, C.c_int32),
                          _p(co, C.c_int32), C.byref(nc), C.byref(sp))
    return ci[:sp.value].copy(), co[:nc.value + 1].copy()


def build_and_export_octree(points, xyzwhl, num_levels=3):
    points = _f32(points)
    xyzwhl = _f32(xyzwhl)
    n = points.shape[0]
    num_nodes = sum(8**i for i in range(num_levels + 1))
    num_leaves = 8**num_levels
    boxes = np.zeros((num_nodes, 6), np.float32)
    pt_inds = np.zeros(n, np.int32)
    pt_start_len = np.zeros((num_leaves, 2), np.int32)
    lib().orc_build_and_export_octree(_p(points, C.c_float), _p(xyzwhl, C.c_float), n, num_levels,
                                      _p(boxes, C.c_float), _p(pt_inds, C.c_int32),
                                      _p(pt_start_len, C.c_int32))
    return boxes, pt_inds, pt_start_len


def octree_ball_query(coords, mean_active, radius):
    """functions.py:14-44 (host glue + retry loop)."""
    coords = _f32(coords)
    xyz_max = coords.max(0)
    xyz_min = coords.min(0)
    xyzwhl = np.concatenate([(xyz_max + xyz_min) / 2, xyz_max - xyz_min]).astype(np.float32)
    n = coords.shape[0]
    boxes, pt_inds, pt_start_len = build_and_export_octree(coords, xyzwhl, 3)
    f = lib().orc_octree_ball_query
    f.restype = C.c_int64
    while True:
        out_inds = np.zeros(n * mean_active, np.int32)
        out_start_len = np.zeros((n, 2), np.int32)
        n_totals = f(_p(coords, C.c_float), _p(boxes, C.c_float), _p(pt_inds, C.c_int32),
                     _p(pt_start_len, C.c_int32), n, int(mean_active), C.c_float(radius),
                     _p(out_inds, C.c_int32), _p(out_start_len, C.c_int32))
        if n_totals <= n * mean_active:
            break
        mean_active = int(n_totals // n + 1)
    return out_inds[:n_totals], out_start_len


def _seg(name, inp, offsets):
    inp = _f32(inp)
    offsets = _i32(offsets)
    nP = offsets.shape[0] - 1
    Cn = inp.shape[1]
    out = np.zeros((nP, Cn), np.float32)
    getattr(lib(), name)(_p(inp, C.c_float), _p(offsets, C.c_int32), nP, Cn, _p(out, C.c_float))
    return out


def sec_mean(inp, offsets):
    return _seg('orc_sec_mean', inp, offsets)


def sec_min(inp, offsets):
    return _seg('orc_sec_min', inp, offsets)


def sec_max(inp, offsets):
    return _seg('orc_sec_max', inp, offsets)


def global_avg_pool(feats, offsets):
    return _seg('orc_global_avg_pool_fp', feats, offsets)


def global_avg_pool_bp(d_out, offsets, sumNPoint):
    d_out = _f32(d_out)
    offsets = _i32(offsets)
    nP, Cn = d_out.shape
    d_feats = np.zeros((sumNPoint, Cn), np.float32)
    lib().orc_global_avg_pool_bp(_p(d_feats, C.c_float), _p(offsets, C.c_int32),
                                 _p(d_out, C.c_float), nP, Cn)
    return d_feats


def get_mask_iou_on_cluster(proposals_idx, proposals_offset, instance_labels, instance_pointnum):
    pi, po = _i32(proposals_idx), _i32(proposals_offset)
    il, ip = _i64(instance_labels), _i32(instance_pointnum)
    nI, nP = ip.shape[0], po.shape[0] - 1
    iou = np.zeros((nP, nI), np.float32)
    lib().orc_get_mask_iou_on_cluster(_p(pi, C.c_int32), _p(po, C.c_int32), _p(il, C.c_int64),
                                      _p(ip, C.c_int32), nI, nP, _p(iou, C.c_float))
    return iou


def get_mask_iou_on_pred(proposals_idx, proposals_offset, instance_labels, instance_pointnum,
                         mask_scores_sigmoid):
    pi, po = _i32(proposals_idx), _i32(proposals_offset)
    il, ip = _i64(instance_labels), _i32(instance_pointnum)
    ms = _f32(mask_scores_sigmoid)
    nI, nP = ip.shape[0], po.shape[0] - 1
    iou = np.zeros((nP, nI), np.float32)
    lib().orc_get_mask_iou_on_pred(_p(pi, C.c_int32), _p(po, C.c_int32), _p(il, C.c_int64),
                                   _p(ip, C.c_int32), _p(ms, C.c_float), nI, nP, _p(iou, C.c_float))
    return iou


def get_mask_label(proposals_idx, proposals_offset, instance_labels, instance_cls,
                   instance_pointnum, proposals_iou, iou_thr):
    pi, po = _i32(proposals_idx), _i32(proposals_offset)
    il, ic = _i64(instance_labels), _i64(instance_cls)
    iou = _f32(proposals_iou)
    nI, nP = ic.shape[0], po.shape[0] - 1
    ml = np.full(pi.shape[0], -1.0, np.float32)
    lib().orc_get_mask_label(_p(pi, C.c_int32), _p(po, C.c_int32), _p(il, C.c_int64),
                             _p(ic, C.c_int64), _p(iou, C.c_float), nI, nP, C.c_float(iou_thr),
                             _p(ml, C.c_float))
    return ml


# ------------------------------------------------------------------ sparse conv
def subm_rulebook(indices, spatial_shape):
    indices = _i32(indices)
    shape = _i32(spatial_shape)
    M = indices.shape[0]
    nbr = np.empty((M, 27), np.int32)
    lib('conv').orc_subm_rulebook(_p(indices, C.c_int32), M, _p(shape, C.c_int32),
                                  _p(nbr, C.c_int32))
    return nbr


def subm_conv3d(feats, nbr, weight):
    """weight [Cout,3,3,3,Cin] -> out [M,Cout]"""
    feats = _f32(feats)
    nbr = _i32(nbr)
    W = _f32(weight)
    M, Cin = feats.shape
    Cout = W.shape[0]
    out = np.empty((M, Cout), np.float32)
    lib('conv').orc_subm_conv3d(_p(feats, C.c_float), _p(nbr, C.c_int32), M, Cin, Cout,
                                _p(W, C.c_float), _p(out, C.c_float))
    return out


def down_rulebook(indices, spatial_shape):
    """-> (out_indices i32[M_out,4], in2out i32[M], child i32[M_out,8], out_shape)"""
    indices = _i32(indices)
    shape = _i32(spatial_shape)
    M = indices.shape[0]
    out_idx = np.zeros((max(M, 1), 4), np.int32)
    in2out = np.zeros(M, np.int32)
    m_out = lib('conv').orc_down_rulebook(_p(indices, C.c_int32), M, _p(shape, C.c_int32),
                                          _p(out_idx, C.c_int32), _p(in2out, C.c_int32))
    out_idx = out_idx[:m_out].copy()
    child = np.empty((m_out, 8), np.int32)
    lib('conv').orc_down_children(_p(indices, C.c_int32), _p(in2out, C.c_int32), M, m_out,
                                  _p(child, C.c_int32))
    return out_idx, in2out, child, [int(s) // 2 for s in shape]


def sparse_conv3d_k2s2(feats, child, weight):
    feats = _f32(feats)
    child = _i32(child)
    W = _f32(weight)
    Cin = feats.shape[1]
    Cout = W.shape[0]
    m_out = child.shape[0]
    out = np.empty((m_out, Cout), np.float32)
    lib('conv').orc_sparse_conv3d_k2s2(_p(feats, C.c_float), _p(child, C.c_int32), m_out, Cin,
                                       Cout, _p(W, C.c_float), _p(out, C.c_float))
    return out


def inverse_conv3d_k2(feats, indices_fine, in2out, weight):
    feats = _f32(feats)
    indices_fine = _i32(indices_fine)
    in2out = _i32(in2out)
    W = _f32(weight)
    M = indices_fine.shape[0]
    Cin = feats.shape[1]
    Cout = W.shape[0]
    out = np.empty((M, Cout), np.float32)
    lib('conv').orc_inverse_conv3d_k2(_p(feats, C.c_float), _p(indices_fine, C.c_int32),
                                      _p(in2out, C.c_int32), M, Cin, Cout, _p(W, C.c_float),
                                      _p(out, C.c_float))
    return out
