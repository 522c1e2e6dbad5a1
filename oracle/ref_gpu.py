"""TEST INFRASTRUCTURE.  ctypes front end of oracle/_ref/sg_ref_gpu_ops.so: the reference's OWN CUDA
kernels (/root/reference/softgroup/ops/src/cuda.cu, unmodified) compiled with hipcc for gfx950 by
oracle/build_ref.py.  Each function below reproduces the allocation / zero-fill / retry protocol of
the reference's Python wrapper (softgroup/ops/functions.py, lines cited per function) around the
reference's host launcher, on torch CUDA tensors.

Only tests/ may import this module (it is the GPU-side checker: HIP kernels == reference kernels,
oracle/sg_oracle.c == reference kernels).  The product (softgroup_amd/) never loads it."""
import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, '_ref', 'sg_ref_gpu_ops.so')
_lib = None


def available():
    return os.path.exists(SO)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(SO)
        for name in ('sgref_ballquery_batch_p', 'sgref_octree_ball_query', 'sgref_sync'):
            getattr(_lib, name).restype = C.c_int
    return _lib


def _p(t):
    return C.c_void_p(t.data_ptr())


def _sync():
    # the reference launches most kernels on the legacy default stream (SURVEY 8b "Threading"):
    # order them after torch's stream and wait for them before torch reads the result
    torch.cuda.synchronize()


def _done():
    rc = lib().sgref_sync()
    assert rc == 0, f'reference kernel failed (hip error {rc})'


def voxelization(feats, map_rule, mode=4):
    """functions.py:200-221 (Voxelization.forward)"""
    assert map_rule.is_contiguous() and feats.is_contiguous()
    N, Cn = feats.size()
    M, mA1 = map_rule.size(0), map_rule.size(1)
    out = torch.zeros((M, Cn), dtype=torch.float32, device=feats.device)
    _sync()
    lib().sgref_voxelize_fp(M, mA1 - 1, Cn, _p(feats), _p(out), _p(map_rule), int(mode == 4))
    _done()
    return out


def voxelization_bp(d_output_feats, map_rule, N, mode=4):
    """functions.py:223-234 (Voxelization.backward)"""
    M, Cn = d_output_feats.size()
    mA1 = map_rule.size(1)
    d_feats = torch.zeros((N, Cn), dtype=torch.float32, device=d_output_feats.device)
    _sync()
    lib().sgref_voxelize_bp(M, mA1 - 1, Cn, _p(d_output_feats.contiguous()), _p(d_feats),
                            _p(map_rule), int(mode == 4))
    _done()
    return d_feats


def ballquery_batch_p(coords, batch_idxs, batch_offsets, radius, mean_active):
    """functions.py:237-275 (BallQueryBatchP.forward incl. the grow-and-retry loop)"""
    n = coords.size(0)
    assert coords.is_contiguous() and batch_idxs.is_contiguous() and batch_offsets.is_contiguous()
    lib().sgref_set_stream(None)
    while True:
        idx = torch.zeros(n * mean_active, dtype=torch.int32, device=coords.device)
        start_len = torch.zeros((n, 2), dtype=torch.int32, device=coords.device)
        _sync()
        n_active = lib().sgref_ballquery_batch_p(n, int(mean_active), C.c_float(radius), _p(coords),
                                                 _p(batch_idxs), _p(batch_offsets), _p(idx),
                                                 _p(start_len))
        _done()
        if n_active <= n * mean_active:
            break
        mean_active = int(n_active // n + 1)
    return idx[:n_active], start_len


def octree_ball_query(coords, boxes, pt_inds, pt_start_len, mean_active, radius):
    """functions.py:30-44 (the GPU half of octree_ball_query; the octree itself comes from the
    reference's CPU build_and_export_octree or its golden export)"""
    n = coords.size(0)
    lib().sgref_set_stream(None)
    while True:
        out_inds = torch.zeros(n * mean_active, dtype=torch.int32, device=coords.device)
        out_start_len = torch.zeros((n, 2), dtype=torch.int32, device=coords.device)
        _sync()
        n_totals = lib().sgref_octree_ball_query(_p(coords), _p(boxes), _p(pt_inds), _p(pt_start_len),
                                                 _p(out_inds), _p(out_start_len), int(mean_active),
                                                 C.c_float(radius), n, boxes.size(0),
                                                 pt_start_len.size(0))
        _done()
        if n_totals <= n * mean_active:
            break
        mean_active = int(n_totals // n + 1)
    return out_inds[:n_totals], out_start_len


def _seg(name, inp, offsets):
    """functions.py:351-438 (SecMean / SecMin / SecMax.forward)"""
    nP = offsets.size(0) - 1
    Cn = inp.size(1)
    assert inp.is_contiguous() and offsets.is_contiguous()
    out = torch.zeros((nP, Cn), dtype=torch.float32, device=inp.device)
    _sync()
    getattr(lib(), name)(nP, Cn, _p(inp), _p(offsets), _p(out))
    _done()
    return out


def sec_mean(inp, offsets):
    return _seg('sgref_sec_mean', inp, offsets)


def sec_min(inp, offsets):
    return _seg('sgref_sec_min', inp, offsets)


def sec_max(inp, offsets):
    return _seg('sgref_sec_max', inp, offsets)


def global_avg_pool(feats, proposals_offset):
    """functions.py:311-331 (GlobalAvgPool.forward)"""
    return _seg('sgref_global_avg_pool_fp', feats, proposals_offset)


def global_avg_pool_bp(d_output_feats, proposals_offset, sumNPoint):
    """functions.py:333-348 (GlobalAvgPool.backward)"""
    nP, Cn = d_output_feats.size()
    d_feats = torch.zeros((sumNPoint, Cn), dtype=torch.float32, device=d_output_feats.device)
    _sync()
    lib().sgref_global_avg_pool_bp(nP, Cn, _p(d_feats), _p(proposals_offset),
                                   _p(d_output_feats.contiguous()))
    _done()
    return d_feats


def get_mask_iou_on_cluster(proposals_idx, proposals_offset, instance_labels, instance_pointnum):
    """functions.py:47-83"""
    nI = instance_pointnum.size(0)
    nP = proposals_offset.size(0) - 1
    iou = torch.zeros((nP, nI), dtype=torch.float32, device=proposals_idx.device)
    _sync()
    lib().sgref_get_mask_iou_on_cluster(nI, nP, _p(proposals_idx), _p(proposals_offset),
                                        _p(instance_labels), _p(instance_pointnum), _p(iou))
    _done()
    return iou


def get_mask_iou_on_pred(proposals_idx, proposals_offset, instance_labels, instance_pointnum,
                         mask_scores_sigmoid):
    """functions.py:86-125"""
    nI = instance_pointnum.size(0)
    nP = proposals_offset.size(0) - 1
    iou = torch.zeros((nP, nI), dtype=torch.float32, device=proposals_idx.device)
    _sync()
    lib().sgref_get_mask_iou_on_pred(nI, nP, _p(proposals_idx), _p(proposals_offset),
                                     _p(instance_labels), _p(instance_pointnum), _p(iou),
                                     _p(mask_scores_sigmoid))
    _done()
    return iou


def get_mask_label(proposals_idx, proposals_offset, instance_labels, instance_cls,
                   instance_pointnum, proposals_iou, iou_thr):
    """functions.py:128-165 (mask_label initialised with -1)"""
    nI = instance_pointnum.size(0)
    nP = proposals_offset.size(0) - 1
    ml = torch.ones(proposals_idx.size(0), dtype=torch.float32, device=proposals_idx.device) * -1.
    _sync()
    lib().sgref_get_mask_label(nI, nP, C.c_float(iou_thr), _p(proposals_idx), _p(proposals_offset),
                               _p(instance_labels), _p(instance_cls), _p(proposals_iou.contiguous()),
                               _p(ml))
    _done()
    return ml
