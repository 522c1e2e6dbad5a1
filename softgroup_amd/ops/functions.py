"""Host-side mirror of the reference operator surface ``softgroup/ops/functions.py``.

Same callables, argument meaning, return conventions and assertion behaviour as the reference
(cited per function; paths relative to the reference repository), implemented on top of the
C ABI of ``libsoftgroup_hip.so``.  There is no CPU/PyTorch fallback for the GPU ops: a missing
library or a failing kernel raises ``SoftGroupHipError``.

Differences that are deliberate (and invisible to results):
  * tensors that the reference requires on the CPU (``bfs_cluster`` inputs, the
    ``voxelization_idx`` input inside the model) may also be CUDA tensors; the result then stays
    on the GPU and no PCIe round trip happens.  CPU in -> CPU out exactly like the reference.
  * ``ballquery_batch_p`` / ``octree_ball_query`` size their output with a counting pass, so the
    reference's grow-and-retry loop (functions.py:34-42, 258-266) never re-launches; the returned
    ``idx`` has exactly ``nActive`` entries either way and the CSR start offsets are ascending
    in point order (the reference's come from an atomic cursor and are arbitrary).
"""
import ctypes as C

import numpy as np
import torch
from torch.autograd import Function

from .. import _lib as L


def _dev(*ts):
    for t in ts:
        if t is not None and t.is_cuda:
            return t.device
    return torch.device('cuda', torch.cuda.current_device())


# ---------------------------------------------------------------------------------------------
# ball query
# ---------------------------------------------------------------------------------------------
def ball_query(coords, batch_idxs, batch_offsets, radius, mean_active, with_octree=False):
    """functions.py:7-11"""
    if with_octree:
        return octree_ball_query(coords, mean_active, radius)
    return ballquery_batch_p(coords, batch_idxs, batch_offsets, radius, mean_active)


def octree_ball_query(coords, mean_active, radius):
    """functions.py:14-44.  Octree export on the host (as the reference), walk on the GPU."""
    lib = L.lib()
    dev = _dev(coords)
    if coords.is_cuda and coords.size(0) > 0:
        # device-resident coordinates: the octree is built where they are (sg_octree_build: same boxes,
        # leaf order and leaf ranges as the host export, no .cpu(), no host thread)
        pts = coords.detach().float().contiguous()
        n = pts.size(0)
        boxes = torch.empty((1 + 8 + 64 + 512, 6), dtype=torch.float32, device=dev)
        pt_inds = torch.empty(n, dtype=torch.int32, device=dev)
        pt_start_len = torch.empty((512, 2), dtype=torch.int32, device=dev)
        nb = lib.sg_octree_build_workspace_bytes(n)
        ws = L.workspace(nb, dev)
        st = L.stream()
        L.check(lib.sg_octree_build(L.ptr(pts), n, L.ptr(boxes), L.ptr(pt_inds), L.ptr(pt_start_len),
                                    L.ptr(ws), ws.numel(), st), 'sg_octree_build')
        start_len = torch.zeros((n, 2), dtype=torch.int32, device=dev)
        L.check(lib.sg_octree_ballquery_count(L.ptr(pts), L.ptr(boxes), L.ptr(pt_inds),
                                              L.ptr(pt_start_len), n, float(radius), L.ptr(start_len),
                                              st), 'sg_octree_ballquery_count')
        n_totals = _scan_start_len(start_len, n, dev)
        out_inds = torch.empty(n_totals, dtype=torch.int32, device=dev)
        L.check(lib.sg_octree_ballquery_fill(L.ptr(pts), L.ptr(boxes), L.ptr(pt_inds),
                                             L.ptr(pt_start_len), n, float(radius), L.ptr(start_len),
                                             L.ptr(out_inds), st), 'sg_octree_ballquery_fill')
        out_inds._sg_flags = LISTS_RADIUS
        return out_inds, start_len
    coords_cpu = coords.detach().cpu().float().contiguous()
    assert coords_cpu.is_contiguous()
    n = coords_cpu.size(0)
    if n == 0:
        return (torch.zeros(0, dtype=torch.int32, device=dev),
                torch.zeros((0, 2), dtype=torch.int32, device=dev))
    xyz_max = coords_cpu.max(0)[0]
    xyz_min = coords_cpu.min(0)[0]
    xyzwhl = torch.cat([(xyz_max + xyz_min) / 2, xyz_max - xyz_min]).contiguous()
    num_levels = 3
    boxes = torch.zeros((1 + 8 + 64 + 512, 6), dtype=torch.float32)
    pt_inds = torch.zeros(n, dtype=torch.int32)
    pt_start_len = torch.zeros((512, 2), dtype=torch.int32)
    L.check(lib.sg_octree_build_host(L.ptr(coords_cpu), L.ptr(xyzwhl), n, num_levels, L.ptr(boxes),
                                     L.ptr(pt_inds), L.ptr(pt_start_len)), 'sg_octree_build_host')
    boxes, pt_inds, pt_start_len = boxes.to(dev), pt_inds.to(dev), pt_start_len.to(dev)
    pts = coords_cpu.to(dev) if not coords.is_cuda else coords.detach().float().contiguous()
    start_len = torch.zeros((n, 2), dtype=torch.int32, device=dev)
    st = L.stream()
    L.check(lib.sg_octree_ballquery_count(L.ptr(pts), L.ptr(boxes), L.ptr(pt_inds),
                                          L.ptr(pt_start_len), n, float(radius), L.ptr(start_len),
                                          st), 'sg_octree_ballquery_count')
    n_totals = _scan_start_len(start_len, n, dev)
    out_inds = torch.empty(n_totals, dtype=torch.int32, device=dev)
    L.check(lib.sg_octree_ballquery_fill(L.ptr(pts), L.ptr(boxes), L.ptr(pt_inds),
                                         L.ptr(pt_start_len), n, float(radius), L.ptr(start_len),
                                         L.ptr(out_inds), st), 'sg_octree_ballquery_fill')
    out_inds._sg_flags = LISTS_RADIUS
    return out_inds, start_len


def _scan_start_len(start_len, n, dev):
    lib = L.lib()
    meta = torch.zeros(2, dtype=torch.int32, device=dev)
    nb = lib.sg_scan_workspace_bytes(n)
    ws = L.workspace(nb, dev)
    L.check(lib.sg_exclusive_scan_startlen(L.ptr(start_len), n, L.ptr(meta), L.ptr(ws), nb,
                                           L.stream()), 'sg_exclusive_scan_startlen')
    return int(meta[0].item())


class BallQueryBatchP(Function):
    """functions.py:237-275"""

    @staticmethod
    def forward(ctx, coords, batch_idxs, batch_offsets, radius, meanActive):
        n = coords.size(0)
        assert coords.is_contiguous() and coords.is_cuda
        assert batch_idxs.is_contiguous() and batch_idxs.is_cuda
        assert batch_offsets.is_contiguous() and batch_offsets.is_cuda
        lib = L.lib()
        dev = coords.device
        coords = coords.detach()
        if coords.dtype != torch.float32:
            coords = coords.float()
        if batch_idxs.dtype != torch.int32:
            batch_idxs = batch_idxs.int()
        start_len = torch.zeros((n, 2), dtype=torch.int32, device=dev)
        if n == 0:
            return torch.zeros(0, dtype=torch.int32, device=dev), start_len
        nb = lib.sg_ballquery_workspace_bytes(n)
        ws = L.workspace(nb, dev)
        st = L.stream()
        L.check(lib.sg_ballquery_build_grid(L.ptr(coords), L.ptr(batch_idxs), n, float(radius),
                                            L.ptr(ws), nb, st), 'sg_ballquery_build_grid')
        L.check(lib.sg_ballquery_count(L.ptr(coords), L.ptr(batch_idxs), n, float(radius),
                                       L.ptr(start_len), None, L.ptr(ws), nb, st),
                'sg_ballquery_count')
        nActive = _scan_start_len(start_len, n, dev)
        idx = torch.empty(nActive, dtype=torch.int32, device=dev)
        L.check(lib.sg_ballquery_fill(L.ptr(coords), L.ptr(batch_idxs), n, float(radius),
                                      L.ptr(start_len), L.ptr(idx), L.ptr(ws), nb, st),
                'sg_ballquery_fill')
        return idx, start_len

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None, None


def ballquery_batch_p(coords, batch_idxs, batch_offsets, radius, meanActive):
    idx, start_len = BallQueryBatchP.apply(coords, batch_idxs, batch_offsets, radius, meanActive)
    idx._sg_flags = LISTS_SORTED | LISTS_RADIUS   # lets bfs_cluster skip its probes
    return idx, start_len


# ---------------------------------------------------------------------------------------------
# clustering
# ---------------------------------------------------------------------------------------------
def _lists_sorted(idx, start_len):
    """True iff every CSR neighbour list is strictly ascending (decides the membership test)."""
    if idx.numel() < 2:
        return True
    bad = idx[1:] <= idx[:-1]
    starts = start_len[:, 0].long()
    lens = start_len[:, 1].long()
    s = starts[(lens > 0) & (starts > 0)] - 1
    bad[s] = False
    return not bool(bad.any().item())


LISTS_SORTED, LISTS_RADIUS = 1, 2


def bfs_cluster_segments(ball_query_idxs, start_len, seg_thr, seg_of_point=None, list_flags=None):
    """Generalised clustering entry: several classes (segments) in one launch.

    ball_query_idxs int32 [nActive], start_len int32 [n,2] (CUDA), seg_thr float32 [n_seg]
    (threshold of every segment, already multiplied by the class mean), seg_of_point int32 [n].
    Returns (cluster_idxs int32 [S,2], cluster_offsets int32 [nC+1]) on the GPU.
    """
    lib = L.lib()
    dev = ball_query_idxs.device
    n = start_len.size(0)
    if list_flags is None:
        list_flags = getattr(ball_query_idxs, '_sg_flags', None)
    if list_flags is None:     # unknown producer: trust nothing except what we can verify
        list_flags = LISTS_SORTED if _lists_sorted(ball_query_idxs, start_len) else 0
    n_edges = ball_query_idxs.numel()
    nb = lib.sg_bfs_workspace_bytes(n, n_edges)
    ws = L.workspace(nb, dev)
    st = L.stream()
    nc, sp = C.c_int32(0), C.c_int32(0)
    L.check(lib.sg_bfs_cluster_label(L.ptr(ball_query_idxs), L.ptr(start_len), n, n_edges,
                                     int(list_flags), L.ptr(seg_of_point), L.ptr(seg_thr),
                                     seg_thr.numel(), C.byref(nc), C.byref(sp), L.ptr(ws), nb, st),
            'sg_bfs_cluster_label')
    cluster_idxs = torch.empty((sp.value, 2), dtype=torch.int32, device=dev)
    cluster_offsets = torch.zeros(nc.value + 1, dtype=torch.int32, device=dev)
    L.check(lib.sg_bfs_cluster_emit(L.ptr(ball_query_idxs), L.ptr(start_len), n, n_edges,
                                    L.ptr(seg_of_point), L.ptr(seg_thr), nc.value, sp.value,
                                    L.ptr(cluster_idxs), L.ptr(cluster_offsets), L.ptr(ws), nb, st),
            'sg_bfs_cluster_emit')
    return cluster_idxs, cluster_offsets


def bfs_cluster(cluster_numpoint_mean, ball_query_idxs, start_len, threshold, class_id):
    """functions.py:278-308.  CPU tensors in -> CPU tensors out (reference contract);
    CUDA tensors in -> CUDA tensors out.  No gradient (like the reference)."""
    assert cluster_numpoint_mean.is_contiguous()
    assert ball_query_idxs.is_contiguous()
    assert start_len.is_contiguous()
    out_dev = ball_query_idxs.device
    dev = _dev(ball_query_idxs, start_len)
    # thr = threshold or threshold * mean, in fp32 (bfs_cluster.cpp:73-79)
    mean = np.float32(cluster_numpoint_mean.detach().cpu().float()[class_id].item())
    thr = np.float32(threshold) if mean == np.float32(-1) else np.float32(threshold) * mean
    seg_thr = torch.tensor([float(thr)], dtype=torch.float32, device=dev)
    flags = getattr(ball_query_idxs, '_sg_flags', None)     # set by our own ball queries
    idxs = ball_query_idxs.detach().to(dev, torch.int32)
    sl = start_len.detach().to(dev, torch.int32)
    cluster_idxs, cluster_offsets = bfs_cluster_segments(idxs, sl, seg_thr, None, flags)
    return cluster_idxs.to(out_dev), cluster_offsets.to(out_dev)


class BFSCluster(Function):
    """name kept for parity with functions.py:278 (``BFSCluster.apply`` == ``bfs_cluster``)"""

    @staticmethod
    def forward(ctx, cluster_numpoint_mean, ball_query_idxs, start_len, threshold, class_id):
        return bfs_cluster(cluster_numpoint_mean, ball_query_idxs, start_len, threshold, class_id)

    @staticmethod
    def backward(ctx, a=None):
        return None


# ---------------------------------------------------------------------------------------------
# voxelisation
# ---------------------------------------------------------------------------------------------
class Voxelization_Idx(Function):
    """functions.py:168-197.  coords long [N,3|4].  CPU input -> host C++ path (fork-safe, no GPU
    context: this is what DataLoader workers call, data/custom.py:239); CUDA input -> device path.
    Returns (output_coords long [M,ncol], input_map int [N], output_map int [M,maxActive+1])."""

    @staticmethod
    def forward(ctx, coords, batchsize, mode=4):
        assert coords.is_contiguous()
        lib = L.lib()
        N, ncol = coords.size(0), coords.size(1)
        coords = coords.long() if coords.dtype != torch.int64 else coords
        if not coords.is_cuda:
            input_map = torch.zeros(N, dtype=torch.int32)
            M, mA = C.c_int32(0), C.c_int32(0)
            L.check(lib.sg_voxelize_idx_host(L.ptr(coords), N, ncol, mode, L.ptr(input_map),
                                             C.byref(M), C.byref(mA)), 'sg_voxelize_idx_host')
            output_coords = torch.zeros((M.value, ncol), dtype=torch.int64)
            output_map = torch.zeros((M.value, mA.value + 1), dtype=torch.int32)
            L.check(lib.sg_voxelize_idx_fill_host(L.ptr(coords), N, ncol, mode, L.ptr(input_map),
                                                  M.value, mA.value, L.ptr(output_coords),
                                                  L.ptr(output_map)), 'sg_voxelize_idx_fill_host')
            return output_coords, input_map, output_map
        dev = coords.device
        input_map = torch.zeros(N, dtype=torch.int32, device=dev)
        meta = torch.zeros(2, dtype=torch.int32, device=dev)
        nb = lib.sg_voxelize_idx_workspace_bytes(N)
        ws = L.workspace(nb, dev)
        st = L.stream()
        L.check(lib.sg_voxelize_idx_build(L.ptr(coords), N, ncol, mode, L.ptr(input_map),
                                          L.ptr(meta), L.ptr(ws), nb, st), 'sg_voxelize_idx_build')
        M, mA = (int(v) for v in meta.tolist())
        output_coords = torch.empty((M, ncol), dtype=torch.int64, device=dev)
        output_map = torch.empty((M, mA + 1), dtype=torch.int32, device=dev)
        L.check(lib.sg_voxelize_idx_fill(L.ptr(coords), N, ncol, mode, L.ptr(input_map), M, mA,
                                         L.ptr(output_coords), L.ptr(output_map), L.ptr(ws), nb, st),
                'sg_voxelize_idx_fill')
        return output_coords, input_map, output_map

    @staticmethod
    def backward(ctx, a=None, b=None, c=None):
        return None, None, None


voxelization_idx = Voxelization_Idx.apply


class Voxelization(Function):
    """functions.py:200-234: feats cuda float [N,C], map_rule cuda int [M,maxActive+1] -> [M,C]."""

    @staticmethod
    def forward(ctx, feats, map_rule, mode=4):
        assert map_rule.is_contiguous()
        assert feats.is_contiguous()
        assert feats.is_cuda and map_rule.is_cuda
        N, Cn = feats.size()
        M = map_rule.size(0)
        maxActive = map_rule.size(1) - 1
        feats32 = feats if feats.dtype == torch.float32 else feats.float()
        output_feats = torch.empty((M, Cn), dtype=torch.float32, device=feats.device)
        ctx.for_backwards = (map_rule, mode, maxActive, N)
        L.check(L.lib().sg_voxelize_fp(L.ptr(feats32), L.ptr(map_rule), M, maxActive, Cn,
                                       int(mode == 4), L.ptr(output_feats), L.stream()),
                'sg_voxelize_fp')
        return output_feats

    @staticmethod
    def backward(ctx, d_output_feats):
        map_rule, mode, maxActive, N = ctx.for_backwards
        M, Cn = d_output_feats.size()
        d_out = d_output_feats.contiguous().float()
        d_feats = torch.zeros((N, Cn), dtype=torch.float32, device=d_out.device)
        L.check(L.lib().sg_voxelize_bp(L.ptr(d_out), L.ptr(map_rule), M, maxActive, Cn,
                                       int(mode == 4), L.ptr(d_feats), L.stream()),
                'sg_voxelize_bp')
        return d_feats, None, None


voxelization = Voxelization.apply


# ---------------------------------------------------------------------------------------------
# segment ops / pooling
# ---------------------------------------------------------------------------------------------
def _seg_call(name, inp, offsets):
    assert inp.is_contiguous()
    assert offsets.is_contiguous()
    nProposal = offsets.size(0) - 1
    Cn = inp.size(1)
    inp32 = inp if inp.dtype == torch.float32 else inp.float()
    off32 = offsets if offsets.dtype == torch.int32 else offsets.int()
    out = torch.zeros((nProposal, Cn), dtype=torch.float32, device=inp.device)
    L.check(getattr(L.lib(), name)(L.ptr(inp32), L.ptr(off32), nProposal, Cn, L.ptr(out),
                                   L.stream()), name)
    return out


class GlobalAvgPool(Function):
    """functions.py:311-348"""

    @staticmethod
    def forward(ctx, feats, proposals_offset):
        sumNPoint, Cn = feats.size()
        out = _seg_call('sg_global_avg_pool_fp', feats, proposals_offset)
        ctx.for_backwards = (proposals_offset, sumNPoint)
        return out

    @staticmethod
    def backward(ctx, d_output_feats):
        nProposal, Cn = d_output_feats.size()
        proposals_offset, sumNPoint = ctx.for_backwards
        d_out = d_output_feats.contiguous().float()
        off32 = proposals_offset if proposals_offset.dtype == torch.int32 else proposals_offset.int()
        d_feats = torch.zeros((sumNPoint, Cn), dtype=torch.float32, device=d_out.device)
        L.check(L.lib().sg_global_avg_pool_bp(L.ptr(d_feats), L.ptr(off32), L.ptr(d_out),
                                              nProposal, Cn, L.stream()), 'sg_global_avg_pool_bp')
        return d_feats, None


global_avg_pool = GlobalAvgPool.apply


class SecMean(Function):
    """functions.py:351-378"""

    @staticmethod
    def forward(ctx, inp, offsets):
        return _seg_call('sg_sec_mean', inp, offsets)

    @staticmethod
    def backward(ctx, a=None):
        return None, None


class SecMin(Function):
    """functions.py:381-408"""

    @staticmethod
    def forward(ctx, inp, offsets):
        return _seg_call('sg_sec_min', inp, offsets)

    @staticmethod
    def backward(ctx, a=None):
        return None, None


class SecMax(Function):
    """functions.py:411-438"""

    @staticmethod
    def forward(ctx, inp, offsets):
        return _seg_call('sg_sec_max', inp, offsets)

    @staticmethod
    def backward(ctx, a=None):
        return None, None


sec_mean = SecMean.apply
sec_min = SecMin.apply
sec_max = SecMax.apply


# ---------------------------------------------------------------------------------------------
# mask IoU / labels (training)
# ---------------------------------------------------------------------------------------------
class GetMaskIoUOnCluster(Function):
    """functions.py:47-83"""

    @staticmethod
    def forward(ctx, proposals_idx, proposals_offset, instance_labels, instance_pointnum):
        nInstance = instance_pointnum.size(0)
        nProposal = proposals_offset.size(0) - 1
        assert proposals_idx.is_contiguous() and proposals_idx.is_cuda
        assert proposals_offset.is_contiguous() and proposals_offset.is_cuda
        assert instance_labels.is_contiguous() and instance_labels.is_cuda
        assert instance_pointnum.is_contiguous() and instance_pointnum.is_cuda
        proposals_iou = torch.zeros((nProposal, nInstance), dtype=torch.float32,
                                    device=proposals_idx.device)
        # converted copies are named locals: they must stay alive until the launch is enqueued
        pidx, poff = proposals_idx.int(), proposals_offset.int()
        ilab, ipn = instance_labels.long(), instance_pointnum.int()
        L.check(L.lib().sg_get_mask_iou_on_cluster(
            L.ptr(pidx), L.ptr(poff), L.ptr(ilab), L.ptr(ipn), nInstance, nProposal,
            L.ptr(proposals_iou), L.stream()), 'sg_get_mask_iou_on_cluster')
        return proposals_iou

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


get_mask_iou_on_cluster = GetMaskIoUOnCluster.apply


class GetMaskIoUOnPred(Function):
    """functions.py:86-125"""

    @staticmethod
    def forward(ctx, proposals_idx, proposals_offset, instance_labels, instance_pointnum,
                mask_scores_sigmoid):
        nInstance = instance_pointnum.size(0)
        nProposal = proposals_offset.size(0) - 1
        assert proposals_idx.is_contiguous() and proposals_idx.is_cuda
        assert proposals_offset.is_contiguous() and proposals_offset.is_cuda
        assert instance_labels.is_contiguous() and instance_labels.is_cuda
        assert instance_pointnum.is_contiguous() and instance_pointnum.is_cuda
        assert mask_scores_sigmoid.is_contiguous() and mask_scores_sigmoid.is_cuda
        proposals_iou = torch.zeros((nProposal, nInstance), dtype=torch.float32,
                                    device=proposals_idx.device)
        pidx, poff = proposals_idx.int(), proposals_offset.int()
        ilab, ipn = instance_labels.long(), instance_pointnum.int()
        sig = mask_scores_sigmoid.float()
        L.check(L.lib().sg_get_mask_iou_on_pred(
            L.ptr(pidx), L.ptr(poff), L.ptr(ilab), L.ptr(ipn), L.ptr(sig), nInstance, nProposal,
            L.ptr(proposals_iou), L.stream()), 'sg_get_mask_iou_on_pred')
        return proposals_iou

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None, None


get_mask_iou_on_pred = GetMaskIoUOnPred.apply


class GetMaskLabel(Function):
    """functions.py:128-165"""

    @staticmethod
    def forward(ctx, proposals_idx, proposals_offset, instance_labels, instance_cls,
                instance_pointnum, proposals_iou, iou_thr):
        nInstance = instance_pointnum.size(0)
        nProposal = proposals_offset.size(0) - 1
        assert proposals_iou.is_contiguous() and proposals_iou.is_cuda
        assert proposals_idx.is_contiguous() and proposals_idx.is_cuda
        assert proposals_offset.is_contiguous() and proposals_offset.is_cuda
        assert instance_labels.is_contiguous() and instance_labels.is_cuda
        assert instance_cls.is_contiguous() and instance_cls.is_cuda
        mask_label = torch.full(proposals_idx.shape, -1.0, dtype=torch.float32,
                                device=proposals_idx.device)
        pidx, poff = proposals_idx.int(), proposals_offset.int()
        ilab, icls = instance_labels.long(), instance_cls.long()
        iou = proposals_iou.float()
        L.check(L.lib().sg_get_mask_label(
            L.ptr(pidx), L.ptr(poff), L.ptr(ilab), L.ptr(icls), L.ptr(iou), nInstance, nProposal,
            float(iou_thr), L.ptr(mask_label), L.stream()), 'sg_get_mask_label')
        return mask_label

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None, None, None, None


get_mask_label = GetMaskLabel.apply


def pyramid_inverse_map(proposals_idx, n_prop, l2p_map, n_voxels):
    """SoftGroup.pyramid_inverse_map (reference softgroup.py:500-507) on the device: proposals over
    level voxels -> proposals over points.  proposals_idx int32 CUDA [S,2] = (proposal, voxel),
    l2p_map int32 CUDA [n] = voxel of every point.  -> (pidx int32 [S',2], poff int32 [n_prop+1])."""
    lib = L.lib()
    dev = proposals_idx.device
    pairs = proposals_idx.int().contiguous()
    l2p = l2p_map.int().contiguous()
    n = l2p.numel()
    out_idx = torch.empty((n, 2), dtype=torch.int32, device=dev)
    out_off = torch.empty(n_prop + 1, dtype=torch.int32, device=dev)
    n_out = torch.zeros(1, dtype=torch.int32, device=dev)
    nb = lib.sg_pyramid_inverse_map_workspace_bytes(n, n_voxels, n_prop)
    ws = L.workspace(nb, dev)
    L.check(lib.sg_pyramid_inverse_map(L.ptr(pairs), pairs.size(0), n_prop, L.ptr(l2p), n, n_voxels,
                                       L.ptr(out_idx), L.ptr(out_off), L.ptr(n_out), L.ptr(ws), ws.numel(),
                                       L.stream()), 'sg_pyramid_inverse_map')
    return out_idx[:int(n_out.item())], out_off
