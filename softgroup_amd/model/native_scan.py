"""Host side of the native scan driver (csrc/scan_exec.hip, include/softgroup_hip.h
``sg_scan_grouping`` / ``sg_scan_instances``): the grouping head + proposal voxelisation
(reference softgroup/model/softgroup.py:411-480,655-709) and the instance extraction (:537-604)
as one C call each.  This module only marshals: descriptors, one grow-only device arena per
(device, stream), one pinned result buffer per host thread, tensor views of the results."""
import ctypes as C
import threading

import numpy as np
import torch

from .. import _lib as L

_ERR_WORKSPACE = -2     # SG_ERR_WORKSPACE


class GroupingCfg(C.Structure):
    _fields_ = [('n_points', C.c_int), ('n_sem_classes', C.c_int), ('n_seg', C.c_int),
                ('seg_class', C.c_void_p), ('seg_thr', C.c_void_p), ('score_thr', C.c_float),
                ('min_npoint', C.c_int), ('radius', C.c_float), ('batch_size', C.c_int),
                ('voxel_scale', C.c_float), ('voxel_shape', C.c_int), ('feat_channels', C.c_int)]


class GroupingPPCfg(C.Structure):
    """sg_grouping_pp_cfg: the SoftGroup++ grouping (pyramid levels / octree query), one class at a time in C"""
    _fields_ = [('base', GroupingCfg), ('with_pyramid', C.c_int), ('with_octree', C.c_int), ('lvl_fusion', C.c_int),
                ('radius', C.c_double), ('base_size', C.c_double)]


class GroupingResult(C.Structure):
    _fields_ = [('n_selected', C.c_int), ('n_neighbours', C.c_int), ('n_proposals', C.c_int),
                ('sum_npoint', C.c_int), ('n_voxels', C.c_int), ('max_active', C.c_int),
                ('proposals_idx', C.c_size_t), ('proposals_offset', C.c_size_t),
                ('voxel_coords', C.c_size_t), ('voxel_offsets', C.c_size_t),
                ('voxel_feats', C.c_size_t), ('point_to_voxel', C.c_size_t),
                ('arena_used', C.c_size_t), ('arena_needed', C.c_size_t),
                ('deferred_classes', C.c_int), ('reserved_', C.c_int)]


class InstancesCfg(C.Structure):
    _fields_ = [('n_proposals', C.c_int), ('n_classes', C.c_int), ('score_stride', C.c_int),
                ('sum_npoint', C.c_int64), ('n_points', C.c_int), ('cls_score_thr', C.c_float),
                ('mask_score_thr', C.c_float), ('min_npoint', C.c_int)]


class InstancesResult(C.Structure):
    _fields_ = [('n_kept', C.c_int), ('off_class', C.c_size_t), ('off_score', C.c_size_t),
                ('off_text', C.c_size_t), ('text_bytes', C.c_size_t), ('host_needed', C.c_size_t),
                ('arena_used', C.c_size_t), ('arena_needed', C.c_size_t),
                ('bits', C.c_size_t), ('label_id', C.c_size_t)]


_arenas = {}            # (tag, device, stream) -> uint8 CUDA tensor, grow-only
_host = threading.local()


def _arena(tag, nbytes, device):
    key = (tag, device, L.stream())
    t = _arenas.get(key)
    if t is None or t.numel() < nbytes:
        if t is not None:
            del _arenas[key], t          # (the old block goes back to the allocator before the new one is asked for)
        _arenas[key] = t = torch.empty(int(nbytes) + int(nbytes) // 4, dtype=torch.uint8, device=device)   # headroom, as unet_exec._get_arena
    return t


def release_stream(raw_stream):
    """drop the arenas of a stream that is being retired (its scans have finished)"""
    for key in [k for k in _arenas if k[2] == raw_stream]:
        del _arenas[key]


def _view(arena, off, dtype, *shape):
    n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
    return arena[off:off + n].view(dtype).view(*shape)


def grouping(cfg, scores, pt_offsets, coords_float, batch_idxs, point_feats):
    """-> None (nothing selected / no proposal) or a dict of tensors.  ``proposals_idx`` and
    ``proposals_offset`` are fresh tensors; the voxel tensors are views of this stream's arena,
    valid until the next call on the same stream."""
    lib = L.lib()
    dev = scores.device
    res = GroupingResult()
    nbytes = max(_arenas.get(('g', dev, L.stream()), torch.empty(0)).numel(), 96 << 20)
    pp = isinstance(cfg, GroupingPPCfg)      # SoftGroup++: sg_scan_grouping_pp, same result layout
    call = lib.sg_scan_grouping_pp if pp else lib.sg_scan_grouping
    for _ in range(8):
        arena = _arena('g', nbytes, dev)
        rc = call(C.byref(cfg), L.ptr(scores), L.ptr(pt_offsets), L.ptr(coords_float),
                  L.ptr(batch_idxs), L.ptr(point_feats), L.ptr(arena), arena.numel(),
                  C.byref(res), L.stream())
        if rc != _ERR_WORKSPACE:
            break
        nbytes = max(int(res.arena_needed), 2 * arena.numel())
    L.check(rc, 'sg_scan_grouping_pp' if pp else 'sg_scan_grouping')
    if pp:
        cfg = cfg.base
    if res.sum_npoint == 0:
        return None
    S, nP, M = res.sum_npoint, res.n_proposals, res.n_voxels
    if cfg.voxel_shape == 0:
        return dict(proposals_idx=_view(arena, res.proposals_idx, torch.int32, S, 2).clone(),
                    proposals_offset=_view(arena, res.proposals_offset, torch.int32, nP + 1).clone(),
                    n_proposals=nP)
    return dict(
        deferred_classes=int(res.deferred_classes),
        proposals_idx=_view(arena, res.proposals_idx, torch.int32, S, 2).clone(),
        proposals_offset=_view(arena, res.proposals_offset, torch.int32, nP + 1).clone(),
        voxel_coords=_view(arena, res.voxel_coords, torch.int32, M, 4),
        voxel_offsets=_view(arena, res.voxel_offsets, torch.int32, nP + 1),
        voxel_feats=_view(arena, res.voxel_feats, torch.float32, M, cfg.feat_channels),
        point_to_voxel=_view(arena, res.point_to_voxel, torch.int32, S),
        n_proposals=nP)


def instances(cfg, proposals_idx, mask_scores, cls_prob, iou_scores, panoptic=None):
    """-> (label_id int32 [n], conf float32 [n], text str, text_off list[int], panoptic_preds) of the
    kept instances, in the reference's order; text/offsets follow sg_rle_format_device's convention.
    ``panoptic`` = dict(semantic_preds int64 CUDA [N], cls_offset, skip_iou, semantic_classes) also
    runs the panoptic fusion on the bit rows still lying in the arena (-> uint32 numpy [N], else None)."""
    lib = L.lib()
    dev = mask_scores.device
    res = InstancesResult()
    hbuf = getattr(_host, 'buf', None)
    if hbuf is None:
        hbuf = _host.buf = torch.empty(8 << 20, dtype=torch.uint8, pin_memory=True)
    nbytes = max(_arenas.get(('i', dev, L.stream()), torch.empty(0)).numel(), 32 << 20)
    for _ in range(8):
        arena = _arena('i', nbytes, dev)
        rc = lib.sg_scan_instances(C.byref(cfg), L.ptr(proposals_idx), L.ptr(mask_scores), L.ptr(cls_prob),
                                   L.ptr(iou_scores), L.ptr(arena), arena.numel(), L.ptr(hbuf), hbuf.numel(),
                                   C.byref(res), L.stream())
        if rc != _ERR_WORKSPACE:
            break
        if res.host_needed > hbuf.numel():
            hbuf = _host.buf = torch.empty(int(res.host_needed) * 2, dtype=torch.uint8, pin_memory=True)
        if res.arena_needed > arena.numel():
            nbytes = max(int(res.arena_needed), 2 * arena.numel())
    L.check(rc, 'sg_scan_instances')
    n = res.n_kept
    if n == 0:
        label, conf, text, text_off = np.zeros(0, np.int32), np.zeros(0, np.float32), '', [0]
    else:
        h = hbuf.numpy()
        text_off = h[:8 * (n + 1)].view(np.int64).tolist()
        label = h[res.off_class:res.off_class + 4 * n].view(np.int32).copy()
        conf = h[res.off_score:res.off_score + 4 * n].view(np.float32).copy()
        text = str(memoryview(h)[res.off_text:res.off_text + res.text_bytes], 'ascii')
    pan = None
    if panoptic is not None:
        pan = _panoptic(lib, arena, res, n, cfg.n_points, conf, panoptic, dev)
    return label, conf, text, text_off, pan


def _panoptic(lib, arena, res, n, n_points, conf, p, dev):
    """sg_panoptic_fusion over the kept instances' bit rows; the visiting order is decided here
    exactly as the reference does (np.argsort of the confidences, reversed; softgroup.py:613)"""
    sem = p['semantic_preds'].contiguous()
    assert sem.dtype == torch.int64 and sem.numel() == n_points
    order = np.argsort(conf)[::-1].astype(np.int32)
    order_d = torch.from_numpy(np.ascontiguousarray(order)).to(dev, non_blocking=False)
    out = torch.empty(n_points, dtype=torch.int32, device=dev)
    nb = lib.sg_panoptic_fusion_workspace_bytes(n, n_points)
    ws = _arena('p', nb, dev)
    bits = arena.data_ptr() + res.bits if n else None
    labels = arena.data_ptr() + res.label_id if n else None
    L.check(lib.sg_panoptic_fusion(bits, n, n_points, L.ptr(order_d), labels, L.ptr(sem), int(p['cls_offset']),
                                   float(p['skip_iou']), int(p['semantic_classes']), int(p.get('thing_class_min', 11)),
                                   L.ptr(out), L.ptr(ws), ws.numel(), L.stream()), 'sg_panoptic_fusion')
    return out.cpu().numpy().view(np.uint32)
