"""Host side of ``sg_scan_forward`` (csrc/scan_forward.hip, include/softgroup_hip.h): one scan of
``SoftGroup.forward_test`` (reference softgroup/model/softgroup.py:299-361) as ONE C call.  This module
only marshals: the descriptor (cached per weight state), one grow-only device arena per (device,
stream), the pinned host blocks, and the result dict the reference returns.  ctypes releases the
interpreter lock for the duration of the call, so scan threads (``model.scan_contexts``) do not contend
for it while their scans run (reference loop: tools/test.py:145-150, one scan per process at a time)."""
import ctypes as C
import os
import threading
import weakref

import numpy as np
import torch
from torch import nn

from .. import _lib as L
from ..spconv import core as spcore
from ..spconv import unet_exec as UE
from . import native_scan as NS

_ERR_WORKSPACE = -2
DENSE_MAX = 16
ENABLED = os.environ.get('SG_SCAN_FORWARD', '1') != '0'      # developer A/B knob: 0 = the staged path of round 5


class Mlp2(C.Structure):          # sg_mlp2
    _fields_ = [('w1', C.c_void_p), ('b1', C.c_void_p), ('bn_scale', C.c_void_p), ('bn_shift', C.c_void_p),
                ('w2', C.c_void_p), ('b2', C.c_void_p), ('out', C.c_int)]


class Linear(C.Structure):        # sg_linear
    _fields_ = [('w', C.c_void_p), ('b', C.c_void_p), ('out', C.c_int), ('in', C.c_int)]


class DenseItem(C.Structure):     # sg_scan_dense_item
    _fields_ = [('kind', C.c_int), ('ptr', C.c_void_p), ('bytes', C.c_size_t)]


class ScanDesc(C.Structure):      # sg_scan_desc
    _fields_ = [('backbone', C.POINTER(UE._Desc)), ('tiny', C.POINTER(UE._Desc)),
                ('semantic', Mlp2), ('offset', Mlp2), ('mask', Mlp2), ('cls', Linear), ('iou', Linear),
                ('channels', C.c_int), ('with_coords', C.c_int), ('semantic_classes', C.c_int),
                ('instance_classes', C.c_int), ('grouping', NS.GroupingCfg), ('cls_score_thr', C.c_float),
                ('mask_score_thr', C.c_float), ('min_npoint', C.c_int), ('want_instances', C.c_int),
                ('with_pyramid', C.c_int), ('with_octree', C.c_int), ('pp_radius', C.c_double),
                ('pp_base_size', C.c_double)]


class ScanInput(C.Structure):     # sg_scan_input
    _fields_ = [('n_points', C.c_int), ('n_voxels', C.c_int), ('max_active', C.c_int), ('batch_size', C.c_int),
                ('spatial_shape', C.c_int * 3), ('feats', C.c_void_p), ('feat_dim', C.c_int),
                ('coords_float', C.c_void_p), ('p2v_map', C.c_void_p), ('v2p_map', C.c_void_p),
                ('v2p_is_int64', C.c_int), ('voxel_coords', C.c_void_p), ('voxel_coords_is_int64', C.c_int),
                ('batch_idxs', C.c_void_p), ('semantic_labels', C.c_void_p), ('instance_labels', C.c_void_p),
                ('n_dense', C.c_int), ('dense', DenseItem * DENSE_MAX)]


class ScanResult(C.Structure):    # sg_scan_result
    _fields_ = [('grouping', NS.GroupingResult), ('instances', NS.InstancesResult),
                ('grouping_base', C.c_size_t), ('instances_base', C.c_size_t),
                ('dense_offset', C.c_size_t * DENSE_MAX), ('dense_bytes', C.c_size_t),
                ('voxel_feats_in', C.c_size_t), ('backbone_out', C.c_size_t), ('output_feats', C.c_size_t),
                ('semantic_scores', C.c_size_t), ('semantic_prob', C.c_size_t), ('pt_offsets', C.c_size_t),
                ('semantic_preds', C.c_size_t), ('tiny_out', C.c_size_t), ('mask_scores', C.c_size_t),
                ('cls_scores', C.c_size_t), ('cls_prob', C.c_size_t), ('iou_scores', C.c_size_t),
                ('arena_used', C.c_size_t), ('arena_needed', C.c_size_t), ('host_dense_needed', C.c_size_t),
                ('host_inst_needed', C.c_size_t), ('stage', C.c_int)]


_arenas = {}            # (device, raw stream) -> uint8 CUDA tensor, grow-only
_host = threading.local()
_desc_lock = threading.Lock()


def release_stream(raw_stream):
    for key in [k for k in _arenas if k[1] == raw_stream]:
        del _arenas[key]


def _arena(nbytes, device):
    key = (device, L.stream())
    t = _arenas.get(key)
    if t is None or t.numel() < nbytes:
        if t is not None:
            del _arenas[key], t
        _arenas[key] = t = torch.empty(int(nbytes) + int(nbytes) // 8, dtype=torch.uint8, device=device)
    return t


def _cfg(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


_IDENT = {}


def _mlp(mlp, c, keep):
    """Linear(c, c), [BatchNorm1d(c) in eval mode,] ReLU, Linear(c, out) (blocks.py:9-27; the mask head has
    no norm: scale 1 / shift 0 -- fma(a, 1, 0) is the identity) -> Mlp2 or None"""
    mods = list(mlp._modules.values())
    if len(mods) == 4 and isinstance(mods[1], nn.BatchNorm1d):
        l1, bn, act, l2 = mods
    elif len(mods) == 3:
        (l1, act, l2), bn = mods, None
    else:
        return None
    if not (isinstance(l1, nn.Linear) and isinstance(act, nn.ReLU) and isinstance(l2, nn.Linear)):
        return None
    if (l1.bias is None or l2.bias is None or tuple(l1.weight.shape) != (c, c) or l2.weight.shape[1] != c
            or l2.weight.shape[0] > 32):
        return None
    if bn is not None and (bn.training or bn.running_mean is None):
        return None
    ts = (l1.weight, l1.bias, l2.weight, l2.bias)
    if not all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in ts):
        return None
    if bn is not None:
        scale, shift = spcore._bn_affine(bn)
    else:
        k = (l1.weight.device, c)
        if k not in _IDENT:
            _IDENT[k] = (torch.ones(c, dtype=torch.float32, device=l1.weight.device),
                         torch.zeros(c, dtype=torch.float32, device=l1.weight.device))
        scale, shift = _IDENT[k]
    keep += [scale, shift]
    d = Mlp2()
    d.w1, d.b1, d.bn_scale, d.bn_shift, d.w2, d.b2 = (t.data_ptr() for t in (l1.weight, l1.bias, scale, shift,
                                                                              l2.weight, l2.bias))
    d.out = l2.weight.shape[0]
    return d


def _lin(lin, c):
    if not (isinstance(lin, nn.Linear) and lin.bias is not None and lin.weight.shape[1] == c and lin.weight.is_cuda
            and lin.weight.dtype == torch.float32 and lin.weight.is_contiguous()):
        return None
    d = Linear()
    d.w, d.b, d.out = lin.weight.data_ptr(), lin.bias.data_ptr(), lin.weight.shape[0]
    setattr(d, 'in', c)
    return d


class ScanForward:
    """descriptor + call for one SoftGroup model (kept in model.__dict__['_scan_forward'])"""

    def __init__(self, model):
        self._model = weakref.ref(model)
        self._key = None
        self._desc = None
        self._keep = []
        self._head_tensors = None

    # ---- eligibility (configuration only; tensors are checked per call)
    def usable(self, model, tasks, lvl_fusion, x4_split):
        if not (getattr(model, 'use_scan_forward', ENABLED) and model.use_native_scan and model.use_executor
                and model.use_fused_heads):
            return False
        if lvl_fusion or x4_split or model.sem2ins_classes or 'panoptic' in tasks:
            return False
        if torch.is_grad_enabled() or torch.is_autocast_enabled('cuda'):
            return False
        g = model.grouping_cfg
        n_seg = model.semantic_classes - len(set(_cfg(g, 'ignore_classes')))
        if not 0 < n_seg <= 32 or _cfg(model.instance_voxel_cfg, 'rand_quantize', False):
            return False
        if _cfg(g, 'with_pyramid', False) or _cfg(g, 'with_octree', False):
            # SoftGroup++: the per-class loop in C (sg_scan_grouping_pp) carries the reference's get_level
            # thresholds; a model whose get_level was replaced keeps the staged path
            if not (model.use_native_grouping_pp and 'get_level' not in model.__dict__
                    and type(model).get_level.__qualname__ == 'SoftGroup.get_level'):
                return False
        return model.channels in (16, 32) and model.semantic_classes <= 32

    def _heads(self, model):
        ts = self._head_tensors
        if ts is None:
            mods = [model.semantic_linear, model.offset_linear]
            if not model.semantic_only:
                mods += [model.mask_linear, model.cls_linear, model.iou_score_linear]
            self._head_owners = [(dct, name) for m in mods for sub in m.modules()
                                 for dct in (sub._parameters, sub._buffers) for name in dct if dct[name] is not None]
            self._head_tensors = True
        return [dct[name] for dct, name in self._head_owners]

    def descriptor(self, model, want_inst):
        """the cached sg_scan_desc, rebuilt when a weight / buffer of the model changed (the executors'
        state keys + the same key over the dense heads), or None when a structure is not covered"""
        bb = model.__dict__.get('_backbone_exec')
        if bb is None:
            bb = model.__dict__['_backbone_exec'] = UE.UNetExecutor(model.unet, model.input_conv, model.output_layer)
        tiny = None
        if want_inst:
            tiny = model.__dict__.get('_tiny_exec')
            if tiny is None:
                tiny = model.__dict__['_tiny_exec'] = UE.UNetExecutor(model.tiny_unet, None, model.tiny_unet_outputlayer)
        if not bb._supported() or (tiny is not None and not tiny._supported()):
            return None
        for ex in (bb, tiny):
            if ex is None:
                continue
            if ex.__dict__.get('_tensor_list') is None:
                ex._tensors()
            for bn in ex._bns:
                if bn.__dict__['training'] or bn._buffers['running_mean'] is None:
                    return None
        d_bb = bb._descriptor()
        d_tiny = tiny._descriptor() if tiny is not None else None
        hs = self._heads(model)
        key = (spcore.cache_epoch(), bool(want_inst), id(d_bb), id(d_tiny), tuple(map(id, hs)),
               tuple(map(UE._VERSION, hs)), tuple(map(UE._DATA_PTR, hs)))
        with _desc_lock:
            if self._desc is not None and key == self._key:
                return self._desc
            keep = [hs, d_bb, d_tiny]
            c = model.channels
            sem, off = _mlp(model.semantic_linear, c, keep), _mlp(model.offset_linear, c, keep)
            if sem is None or off is None or off.out > 4:
                return None
            d = ScanDesc()
            d.backbone = C.pointer(d_bb)
            d.semantic, d.offset = sem, off
            d.channels, d.with_coords = c, int(bool(model.with_coords))
            d.semantic_classes, d.instance_classes = model.semantic_classes, model.instance_classes
            d.want_instances = int(bool(want_inst))
            if want_inst:
                mask = _mlp(model.mask_linear, c, keep)
                cls, iou = _lin(model.cls_linear, c), _lin(model.iou_score_linear, c)
                if mask is None or cls is None or iou is None:
                    return None
                d.tiny = C.pointer(d_tiny)
                d.mask, d.cls, d.iou = mask, cls, iou
                g, v, t = model.grouping_cfg, model.instance_voxel_cfg, model.test_cfg
                _, seg_thr, _, cls32 = model._grouping_constants(hs[0].device)
                keep += [seg_thr, cls32]
                gc = d.grouping
                gc.n_seg, gc.seg_class, gc.seg_thr = cls32.numel(), cls32.data_ptr(), seg_thr.data_ptr()
                gc.score_thr, gc.min_npoint, gc.radius = _cfg(g, 'score_thr'), _cfg(t, 'min_npoint'), _cfg(g, 'radius')
                gc.voxel_scale, gc.voxel_shape = _cfg(v, 'scale'), _cfg(v, 'spatial_shape')
                d.with_pyramid = int(bool(_cfg(g, 'with_pyramid', False)))
                d.with_octree = int(bool(_cfg(g, 'with_octree', False)))
                d.pp_radius, d.pp_base_size = float(_cfg(g, 'radius')), float(_cfg(g, 'pyramid_base_size', 0.02))
                d.cls_score_thr, d.mask_score_thr = _cfg(t, 'cls_score_thr'), _cfg(t, 'mask_score_thr')
                d.min_npoint = _cfg(t, 'min_npoint')
            torch.cuda.current_stream().synchronize()      # (BN affines just made on this stream; once per key)
            self._desc, self._key, self._keep = d, key, keep
            return d

    # ---- run
    def __call__(self, model, batch, tasks, want_inst):
        """-> dict like forward_test's (numpy arrays, the instance list), or None when this path does not
        apply to the batch (the caller takes the staged path)."""
        feats, coords_float = batch['feats'], batch['coords_float']
        p2v, v2p, vc, bidx = batch['p2v_map'], batch['v2p_map'], batch['voxel_coords'], batch['batch_idxs']
        ts = (feats, coords_float, p2v, v2p, vc, bidx)
        if not all(isinstance(t, torch.Tensor) and t.is_cuda for t in ts):
            return None
        if feats.dtype != torch.float32 or coords_float.dtype != torch.float32 or p2v.dtype != torch.int32:
            return None
        if v2p.dtype not in (torch.int32, torch.int64) or vc.dtype not in (torch.int32, torch.int64):
            return None
        d = self.descriptor(model, want_inst)
        if d is None:
            return None
        lib = L.lib()
        dev = feats.device
        feats, coords_float, p2v, v2p, vc = (t.contiguous() for t in (feats, coords_float, p2v, v2p, vc))
        bidx = bidx.int().contiguous() if bidx.dtype != torch.int32 else bidx.contiguous()
        inp = ScanInput()
        N, M = feats.shape[0], vc.shape[0]
        inp.n_points, inp.n_voxels, inp.max_active = N, M, p2v.shape[1] - 1
        inp.batch_size = int(batch['batch_size'])
        inp.spatial_shape = (C.c_int * 3)(*[int(s) for s in batch['spatial_shape']])
        inp.feats, inp.feat_dim, inp.coords_float = feats.data_ptr(), feats.shape[1], coords_float.data_ptr()
        inp.p2v_map, inp.v2p_map, inp.v2p_is_int64 = p2v.data_ptr(), v2p.data_ptr(), int(v2p.dtype == torch.int64)
        inp.voxel_coords, inp.voxel_coords_is_int64 = vc.data_ptr(), int(vc.dtype == torch.int64)
        inp.batch_idxs = bidx.data_ptr()
        # ---- dense results (forward_test's dense_results(): get_point_wise_results + labels + gt_instances)
        items, keep = [], [feats, coords_float, p2v, v2p, vc, bidx]

        def through(name, t, np_dtype=None):
            t = t.contiguous()
            keep.append(t)
            items.append((name, 0, t.data_ptr(), t.numel() * t.element_size(),
                          np_dtype or UE_NP[t.dtype], tuple(t.shape)))

        sem_l, inst_l = batch.get('semantic_labels'), batch.get('instance_labels')
        if 'semantic' in tasks or 'panoptic' in tasks:
            through('semantic_labels', sem_l)
            through('instance_labels', inst_l)
        if 'semantic' in tasks:
            through('coords_float', coords_float)
            through('color_feats', feats)
            items.append(('semantic_preds', 1, None, N * 8, np.int64, (N, )))
            items.append(('offset_preds', 2, None, N * 12, np.float32, (N, 3)))
            through('offset_labels', batch['pt_offset_labels'])
        if want_inst and 'instance' in tasks:
            if not (sem_l.dtype == torch.int64 and inst_l.dtype == torch.int64):
                return None
            sl, il = sem_l.contiguous(), inst_l.contiguous()
            keep += [sl, il]
            inp.semantic_labels, inp.instance_labels = sl.data_ptr(), il.data_ptr()
            items.append(('gt_instances', 3, None, N * 8, np.int64, (N, )))
        assert len(items) <= DENSE_MAX
        inp.n_dense = len(items)
        dense_total = 0
        for i, (_, kind, ptr, nb, _, _) in enumerate(items):
            inp.dense[i].kind, inp.dense[i].ptr, inp.dense[i].bytes = kind, ptr, nb
            dense_total += (nb + 255) // 256 * 256
        stage = torch.empty(max(dense_total, 1), dtype=torch.uint8, pin_memory=True)
        hbuf = getattr(_host, 'buf', None)
        if hbuf is None:
            hbuf = _host.buf = torch.empty(8 << 20, dtype=torch.uint8, pin_memory=True)
        res = ScanResult()
        key = (dev, L.stream())
        cur = _arenas.get(key)
        nbytes = max(cur.numel() if cur is not None else 0, int(lib.sg_scan_arena_bytes(C.byref(d), N, M)))
        rc = 0
        for _ in range(8):
            arena = _arena(nbytes, dev)
            rc = lib.sg_scan_forward(C.byref(d), C.byref(inp), arena.data_ptr(), arena.numel(), stage.data_ptr(),
                                     stage.numel(), hbuf.data_ptr(), hbuf.numel(), C.byref(res), L.stream())
            if rc != _ERR_WORKSPACE:
                break
            if res.host_inst_needed > hbuf.numel():
                hbuf = _host.buf = torch.empty(int(res.host_inst_needed) * 2, dtype=torch.uint8, pin_memory=True)
            if res.arena_needed > arena.numel():
                nbytes = max(int(res.arena_needed), arena.numel() + arena.numel() // 2)
        L.check(rc, 'sg_scan_forward')
        del keep
        # ---- host objects
        out = {}
        from ..util.cast import adopt_pinned_block
        block = adopt_pinned_block(stage)
        for i, (name, _, _, nb, npdt, shape) in enumerate(items):
            o = res.dense_offset[i]
            out[name] = block[o:o + nb].view(npdt).reshape(shape)
        if want_inst and 'instance' in tasks:
            out['pred_instances'] = self._instances(batch['scan_ids'][0], res, hbuf, N)
        self.last = res
        return out

    @staticmethod
    def _instances(scan_id, res, hbuf, n_points):
        r = res.instances
        n = r.n_kept if res.stage >= 4 else 0
        if n == 0:
            return []
        h = hbuf.numpy()
        off = h[:8 * (n + 1)].view(np.int64).tolist()
        label = h[r.off_class:r.off_class + 4 * n].view(np.int32).astype(np.int64)
        conf = h[r.off_score:r.off_score + 4 * n].view(np.float32).copy()
        text = str(memoryview(h)[r.off_text:r.off_text + r.text_bytes], 'ascii')
        n_points = int(n_points)
        return [dict(scan_id=scan_id, label_id=label[k], conf=conf[k],
                     pred_mask=dict(length=n_points, counts=text[off[k]:max(off[k + 1] - 1, off[k])]))
                for k in range(n)]


UE_NP = {torch.float32: np.float32, torch.float64: np.float64, torch.float16: np.float16, torch.int64: np.int64,
         torch.int32: np.int32, torch.int16: np.int16, torch.int8: np.int8, torch.uint8: np.uint8, torch.bool: np.bool_}
