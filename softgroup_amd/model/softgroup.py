"""SoftGroup ``nn.Module`` hosted on the MI355X-native operators.

The constructor signature, attribute/parameter names (checkpoint contract), the
``model(batch, return_loss)`` entry and the result dictionaries are those of the reference
(softgroup/model/softgroup.py); YAML ``model:`` sections are passed unchanged as kwargs.
What differs is how the forward is driven:

  * the backbone runs BN+ReLU+conv(+residual) as single HIP kernels (softgroup_amd.spconv);
  * grouping handles ALL semantic classes in one ball-query + one clustering launch set on the
    GPU (the reference loops over classes with a device->host copy and a single-threaded CPU BFS
    per class, softgroup.py:433-473) -- proposals come out in the same order, bit for bit;
  * proposal voxelisation builds its index on the GPU (reference: CPU hash, softgroup.py:703);
  * instance masks are run-length encoded from sorted (proposal, point) pairs instead of dense
    [nProposal, N] int masks (reference softgroup.py:568-603).
"""
import functools
import ctypes
import os
import threading
import time
from collections import OrderedDict

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..spconv import pytorch as spconv
from ..util import cuda_cast, force_fp32, to_host, to_host_begin, to_host_end, rle_decode, rle_encode_many, rle_encode_runs, rle_text_to_dicts
from ..util.lazy import LazyResults, worker as lazy_worker
from ..spconv.unet_exec import UNetExecutor
from ..spconv.unet_train import UNetTrainExecutor
from .blocks import MLP, ResidualBlock, UBlock


class _Mlp2(ctypes.Structure):       # sg_mlp2 (include/softgroup_hip.h)
    _fields_ = [('w1', ctypes.c_void_p), ('b1', ctypes.c_void_p), ('bn_scale', ctypes.c_void_p),
                ('bn_shift', ctypes.c_void_p), ('w2', ctypes.c_void_p), ('b2', ctypes.c_void_p),
                ('out', ctypes.c_int)]


_scan_local = threading.local()      # per worker thread: its HIP stream
_EARLY_COPY = os.environ.get('SG_EARLY_COPY', '1') != '0'      # (developer A/B knob)
_POOL_LOCK = threading.Lock()        # creation / retirement of a model's scan pool
_PARKED_STREAMS = {}                 # device -> library-owned worker streams of retired pools, for the next pool
# SG_SCAN_PRIO (developer knob): dispatch priority of the scan workers' streams in creation order, one letter
# each -- h(igh) / n(ormal) / l(ow), e.g. "hnnll"; unset: all default
_SCAN_PRIO = [{'h': 1, 'n': 0, 'l': -1}[c] for c in os.environ.get('SG_SCAN_PRIO', '') if c in 'hnl']



def _retire_pool(pool, wait=True):
    """shut a scan pool down and give back what its worker streams held: this module's and the
    executor's per-stream arenas (>= 130 MB per worker) and the library's per-stream runtime state
    (sg_stream_release).  wait=False -- called from one of the pool's own threads -- only stops the
    pool: its streams may still be running that very scan, their state stays until the process ends."""
    pool.shutdown(wait=wait)
    if not wait:
        return
    from . import native_scan as NS
    from . import scan_forward as SF
    from ..spconv import unet_exec as UE
    from .. import _lib as L
    for dev, st in getattr(pool, '_sg_streams', ()):
        with torch.cuda.device(dev):
            st.synchronize()
            raw = st.cuda_stream
            NS.release_stream(raw)
            SF.release_stream(raw)
            UE.release_stream(raw)
            # the workers' streams are the library's own (sg_stream_create): no other Stream object holds this
            # handle, so releasing its per-stream state cannot pull buffers from under anybody else's kernels
            # (ADVICE r5).  The stream itself is PARKED for the next pool's workers, not destroyed: PyTorch's
            # caching allocator keeps blocks and events tied to every stream it has seen and aborts the process
            # when one of them is gone (measured: the next allocation after a hipStreamDestroy).
            L.check(L.lib().sg_stream_release(raw), 'sg_stream_release')
            with _POOL_LOCK:
                _PARKED_STREAMS.setdefault(dev, []).append(st)
    pool._sg_streams = []



def _cfg(cfg, key, default=None):
    """config sections arrive as Munch / dict / namespace depending on the caller"""
    if cfg is None:
        return default
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


class SoftGroup(nn.Module):

    def __init__(self,
                 in_channels=3,
                 channels=32,
                 num_blocks=7,
                 semantic_only=False,
                 semantic_classes=20,
                 instance_classes=18,
                 semantic_weight=None,
                 sem2ins_classes=[],
                 ignore_label=-100,
                 with_coords=True,
                 grouping_cfg=None,
                 instance_voxel_cfg=None,
                 train_cfg=None,
                 test_cfg=None,
                 fixed_modules=[]):
        super().__init__()
        self.in_channels = in_channels + (3 if with_coords else 0)
        self.channels = channels
        self.num_blocks = num_blocks
        self.semantic_only = semantic_only
        self.semantic_classes = semantic_classes
        self.instance_classes = instance_classes
        self.semantic_weight = semantic_weight
        self.sem2ins_classes = sem2ins_classes
        self.ignore_label = ignore_label
        self.with_coords = with_coords
        self.grouping_cfg = grouping_cfg
        self.instance_voxel_cfg = instance_voxel_cfg
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg
        self.fixed_modules = fixed_modules
        self.use_executor = True     # native U-Net executor for inference (same kernels as the modules)
        # train() mode U-Nets as one autograd node each (csrc/unet_train.hip); SG_TRAIN_EXEC=0: the modules
        self.use_train_executor = os.environ.get('SG_TRAIN_EXEC', '1') != '0'
        self.scan_result_hook = None      # callable(result dict) run by the scan worker before the result is handed back (scan_contexts > 1)
        self.use_fused_heads = os.environ.get('SG_FUSED_HEADS', '1') != '0'   # devoxelize + point-wise heads + arg-max as one kernel (inference)
        self.use_scan_forward = os.environ.get('SG_SCAN_FORWARD', '1') != '0'   # the whole scan as ONE C call (csrc/scan_forward.hip)
        self.use_native_grouping_pp = os.environ.get('SG_NATIVE_GROUPING_PP', '1') != '0'   # SoftGroup++ grouping as one C call (sg_scan_grouping_pp)
        self.use_native_scan = os.environ.get('SG_NATIVE_SCAN', '1') != '0'   # grouping head + proposal voxelisation + instance extraction as
        #                              two C calls (csrc/scan_exec.hip) where the configuration allows
        self.async_results = True    # host-side result formatting overlaps the next forward
        self.scan_contexts = 1       # > 1: model(batch) hands the scan to one of that many worker
        #                              threads (own HIP stream each) and returns at once -- several
        #                              scans in flight keep the GPU busy across the host round trips
        #                              and the latency-bound kernels of a single scan

        norm_fn = functools.partial(nn.BatchNorm1d, eps=1e-4, momentum=0.1)

        # sparse U-Net backbone
        self.input_conv = spconv.SparseSequential(
            spconv.SubMConv3d(self.in_channels, channels, kernel_size=3, padding=1, bias=False,
                              indice_key='subm1'))
        self.unet = UBlock([channels * (i + 1) for i in range(num_blocks)], norm_fn, 2,
                           ResidualBlock, indice_key_id=1)
        self.output_layer = spconv.SparseSequential(norm_fn(channels), nn.ReLU())

        # point-wise heads
        self.semantic_linear = MLP(channels, semantic_classes, norm_fn=norm_fn, num_layers=2)
        self.offset_linear = MLP(channels, 3, norm_fn=norm_fn, num_layers=2)

        # top-down refinement
        if not semantic_only:
            self.tiny_unet = UBlock([channels, 2 * channels], norm_fn, 2, ResidualBlock,
                                    indice_key_id=11)
            self.tiny_unet_outputlayer = spconv.SparseSequential(norm_fn(channels), nn.ReLU())
            self.cls_linear = nn.Linear(channels, instance_classes + 1)
            self.mask_linear = MLP(channels, instance_classes + 1, norm_fn=None, num_layers=2)
            self.iou_score_linear = nn.Linear(channels, instance_classes + 1)

        self.init_weights()
        for name in fixed_modules:
            for p in getattr(self, name).parameters():
                p.requires_grad = False

    # ------------------------------------------------------------------ housekeeping
    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm1d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, MLP):
                m.init_weights()
        if not self.semantic_only:
            for lin in (self.cls_linear, self.iou_score_linear):
                nn.init.normal_(lin.weight, 0, 0.01)
                nn.init.constant_(lin.bias, 0)

    # ---- derived state (packed weights, BatchNorm affines, native-executor descriptors, the
    #      results stream) is rebuilt on demand and never copied / pickled with the module
    _DERIVED = ('_scan_forward', '_backbone_exec', '_tiny_exec', '_backbone_train_exec', '_tiny_train_exec', '_results_stream',
                '_grouping_const', '_scan_pool')

    def invalidate_caches(self):
        """Call after writing parameters/buffers through ``tensor.data`` (EMA copies, custom
        initialisers): such writes do not bump the tensors' version counters, so the cached
        packed weights / BatchNorm affines would otherwise go stale.  load_state_dict, .to(),
        train()/eval() and optimizer steps are covered without it."""
        # the pool leaves the module first (a concurrent model(batch) then builds a new one instead of
        # submitting to a pool that is shutting down); scans still in flight must not see
        # half-replaced derived state, so they are waited for -- unless this IS one of the pool's
        # threads (a hook inside forward_test): joining itself would never return
        with _POOL_LOCK:
            pool = self.__dict__.pop('_scan_pool', None)
        if pool is not None:
            own = threading.current_thread() in getattr(pool, '_threads', ())
            _retire_pool(pool, wait=not own)
        spconv.invalidate_caches()
        for k in self._DERIVED:
            self.__dict__.pop(k, None)

    def __getstate__(self):
        state = self.__dict__.copy()
        for k in self._DERIVED:
            state.pop(k, None)
        return state

    def _apply(self, fn, *args, **kwargs):
        self.invalidate_caches()
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self.invalidate_caches()
        return super().load_state_dict(*args, **kwargs)

    def train(self, mode=True):
        self.invalidate_caches()
        super().train(mode)
        for name in self.fixed_modules:      # frozen parts keep BN statistics frozen too
            for m in getattr(self, name).modules():
                if isinstance(m, nn.BatchNorm1d):
                    m.eval()
        return self

    def forward(self, batch, return_loss=False):
        if return_loss:
            return self.forward_train(**batch)
        if self.scan_contexts > 1 and torch.cuda.is_available() and not torch.is_grad_enabled():
            return self._submit_scan(batch)
        return self.forward_test(**batch)

    def _submit_scan(self, batch):
        """Run forward_test for this batch on a worker thread with its own stream; the returned
        dict resolves (waits) on first access, like the lazily formatted results."""
        dev = torch.cuda.current_device()
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())        # the batch tensors are ready from here on
        local = _scan_local

        def job():
            with torch.cuda.device(dev):
                st = getattr(local, 'stream', None)
                if st is None:
                    with _POOL_LOCK:
                        parked = _PARKED_STREAMS.get(dev)
                        st = parked.pop() if parked else None
                    if st is None:
                        from .. import _lib as L_
                        raw = ctypes.c_void_p(0)
                        prio = _SCAN_PRIO[min(len(my_pool._sg_streams), len(_SCAN_PRIO) - 1)] if _SCAN_PRIO else None
                        if prio is None:
                            L_.check(L_.lib().sg_stream_create(ctypes.byref(raw)), 'sg_stream_create')
                        else:
                            L_.check(L_.lib().sg_stream_create_priority(ctypes.byref(raw), prio),
                                     'sg_stream_create_priority')
                        st = torch.cuda.ExternalStream(raw.value, device=dev)
                    local.stream = st
                    my_pool._sg_streams.append((dev, st))      # state released, stream parked when the pool is retired
                with torch.cuda.stream(st), torch.no_grad():
                    st.wait_event(ready)
                    out = self.forward_test(**batch, _inline_results=True)
                    st.synchronize()
            out = dict(out)
            hook = self.scan_result_hook
            if hook is not None:       # on the worker: a consumer-side pass over the results (a digest, a
                hook(out)              # metric) does not have to win the interpreter lock from the workers
            return out

        with _POOL_LOCK:
            pool = self.__dict__.get('_scan_pool')
            if pool is None or pool._max_workers != self.scan_contexts:
                from concurrent.futures import ThreadPoolExecutor
                old = pool
                pool = self.__dict__['_scan_pool'] = ThreadPoolExecutor(
                    max_workers=self.scan_contexts, thread_name_prefix='softgroup-scan')
                pool._sg_streams = []
                if old is not None:      # scan_contexts changed: the old workers finish their scans and go
                    threading.Thread(target=_retire_pool, args=(old, True), daemon=True).start()
            my_pool = pool
            fut = pool.submit(job)       # (under the lock: the pool cannot be retired in between)
        ret = LazyResults(scan_id=batch['scan_ids'][0])
        ret.defer(fut)
        return ret

    # ------------------------------------------------------------------ inference
    @cuda_cast
    def forward_test(self, batch_idxs, voxel_coords, p2v_map, v2p_map, coords_float, feats,
                     semantic_labels, instance_labels, pt_offset_labels, spatial_shape, batch_size,
                     scan_ids, _inline_results=False, **kwargs):
        tcfg = self.test_cfg
        # ---- the whole scan as ONE C call (csrc/scan_forward.hip) where the configuration allows: plain
        #      SoftGroup grouping, fp32 inference, no panoptic fusion; anything else takes the stages below
        tasks = _cfg(tcfg, 'eval_tasks')
        if feats.is_cuda and 'panoptic' not in tasks:
            sf = self.__dict__.get('_scan_forward')
            if sf is None:
                from .scan_forward import ScanForward
                sf = self.__dict__['_scan_forward'] = ScanForward(self)
            if sf.usable(self, tasks, _cfg(tcfg, 'lvl_fusion', False), _cfg(tcfg, 'x4_split', False)):
                want = not self.semantic_only and 'instance' in tasks
                out = sf(self, dict(feats=feats, coords_float=coords_float, p2v_map=p2v_map, v2p_map=v2p_map,
                                    voxel_coords=voxel_coords, batch_idxs=batch_idxs, semantic_labels=semantic_labels,
                                    instance_labels=instance_labels, pt_offset_labels=pt_offset_labels,
                                    spatial_shape=spatial_shape, batch_size=batch_size, scan_ids=scan_ids),
                         tasks, want)
                if out is not None:
                    ret = LazyResults(scan_id=scan_ids[0])
                    ret.update(out)
                    return ret
        color_feats = feats
        if self.with_coords:
            feats = torch.cat((feats, coords_float), 1)
        voxel_feats = ops.voxelization(feats, p2v_map)
        x = spconv.SparseConvTensor(voxel_feats, voxel_coords.int(), spatial_shape, batch_size)

        lvl_fusion = _cfg(tcfg, 'lvl_fusion', False)
        x4_split = _cfg(tcfg, 'x4_split', False)
        semantic_scores, pt_offsets, output_feats, semantic_preds = self._forward_backbone(
            x, v2p_map, x4_split=x4_split, lvl_fusion=lvl_fusion)
        if x4_split:
            coords_float = self.merge_4_parts(coords_float)
            semantic_labels = self.merge_4_parts(semantic_labels)
            instance_labels = self.merge_4_parts(instance_labels)
            pt_offset_labels = self.merge_4_parts(pt_offset_labels)
        if semantic_preds is None:
            semantic_preds = semantic_scores.max(1)[1]
        tasks = _cfg(tcfg, 'eval_tasks')
        ret = LazyResults(scan_id=scan_ids[0])
        inst = None
        want_inst = not self.semantic_only and ('instance' in tasks or 'panoptic' in tasks)

        # every dense per-point result goes to the host in one pinned block.  They are all known
        # once the point-wise heads have run: the copy starts HERE, on a side stream, and crosses
        # PCIe while the grouping head and the refinement run (12 MB, ~0.3 ms that used to sit at
        # the end of the scan)
        def dense_results():
            dense = {}
            if 'semantic' in tasks or 'panoptic' in tasks:
                dense.update(semantic_labels=semantic_labels, instance_labels=instance_labels)
            if 'semantic' in tasks:
                dense.update(self.get_point_wise_results(coords_float, color_feats, semantic_preds,
                                                         pt_offsets, pt_offset_labels, v2p_map,
                                                         lvl_fusion, _device=True))
            if want_inst and 'instance' in tasks:
                dense.update(gt_instances=self.get_gt_instances(semantic_labels, instance_labels,
                                                                _device=True))
            return dense

        early = None
        # (one scan at a time only: with several scans in flight the copy overlaps the OTHER scans anyway,
        # and a second stream per worker measured slightly slower)
        if semantic_scores.is_cuda and not lvl_fusion and _EARLY_COPY and not _inline_results:
            st = getattr(_scan_local, 'copy_stream', None)
            if st is None:
                st = _scan_local.copy_stream = torch.cuda.Stream()
            early = to_host_begin(dense_results(), st)
        if want_inst:
            if lvl_fusion:
                batch_idxs = x.indices[:, 0].int()
                coords_float = ops.voxelization(coords_float, p2v_map)
            inst = None
            if self._native_scan_usable(semantic_scores, output_feats, lvl_fusion, x4_split):
                inst = self._native_grouping_and_refinement(semantic_scores, pt_offsets, batch_idxs,
                                                            coords_float, output_feats, batch_size)
            if inst is None:
                proposals_idx, proposals_offset = self.forward_grouping(
                    semantic_scores, pt_offsets, batch_idxs, coords_float, self.grouping_cfg,
                    lvl_fusion=lvl_fusion, batch_size=None if x4_split else batch_size)
                inst_feats, inst_map = self.clusters_voxelization(
                    proposals_idx, proposals_offset, output_feats, coords_float,
                    **self.instance_voxel_cfg)
                _, cls_scores, iou_scores, mask_scores = self.forward_instance(inst_feats, inst_map)
                inst = (proposals_idx, cls_scores, iou_scores, mask_scores)

        # ---- everything below only turns device results into host objects (numpy arrays, RLE
        #      strings): it can run on the results thread, on its own stream, while the caller
        #      already enqueues the next scan
        def finish():
            out = {}
            out.update(to_host_end(early) if early is not None else to_host(dense_results()))
            if inst is not None:
                # panoptic fusion runs on the device, on the instances' bit rows, where the native
                # instance extraction applies; else on the host over the RLE strings
                fuse_native = ('panoptic' in tasks and self.use_native_scan and not lvl_fusion
                               and not self.sem2ins_classes and inst[1].is_cuda and inst[0].size(0) > 0)
                pred_instances = self.get_instances(scan_ids[0], inst[0], semantic_scores, inst[1],
                                                    inst[2], inst[3], v2p_map=v2p_map,
                                                    lvl_fusion=lvl_fusion,
                                                    _panoptic_sem=semantic_preds if fuse_native else None)
                fused = None
                if fuse_native:
                    pred_instances, fused = pred_instances
                if 'instance' in tasks:
                    out.update(pred_instances=pred_instances)
                if 'panoptic' in tasks:
                    if fused is None:
                        fused = self.panoptic_fusion(semantic_preds.cpu().numpy(), pred_instances)
                    out.update(panoptic_preds=fused)
            return out

        if _inline_results or not (self.async_results and semantic_scores.is_cuda):
            ret.update(finish())
            return ret
        main = torch.cuda.current_stream()
        done = torch.cuda.Event()
        done.record(main)
        side = self.__dict__.get('_results_stream')
        if side is None:
            side = self.__dict__['_results_stream'] = torch.cuda.Stream()
        dev = semantic_scores.device

        def job():
            with torch.cuda.device(dev), torch.cuda.stream(side), torch.no_grad():
                side.wait_event(done)
                out = finish()
                side.synchronize()
            return out

        ret.defer(lazy_worker().submit(job))
        return ret

    # ------------------------------------------------------------------ native scan driver
    def _native_scan_usable(self, semantic_scores, output_feats, lvl_fusion, x4_split):
        g = self.grouping_cfg
        n_seg = self.semantic_classes - len(set(_cfg(g, 'ignore_classes')))
        return (self.use_native_scan and self.use_executor and semantic_scores.is_cuda
                and output_feats.dtype == torch.float32 and not lvl_fusion and not x4_split
                and not self.sem2ins_classes and 0 < n_seg <= 32
                and (not (_cfg(g, 'with_pyramid', False) or _cfg(g, 'with_octree', False))
                     # SoftGroup++ in C (sg_scan_grouping_pp) carries the reference's get_level thresholds: a model
                     # whose get_level was replaced (tests force level 2 on small scenes) keeps the per-class loop
                     or (self.use_native_grouping_pp and 'get_level' not in self.__dict__
                         and type(self).get_level is SoftGroup.get_level))
                and not _cfg(self.instance_voxel_cfg, 'rand_quantize', False))

    def _grouping_constants(self, dev):
        """(class ids, per-class cluster-size thresholds, dummy offsets) on the device, once per
        configuration: thr = npoint_thr (absolute) if the class mean is -1 else npoint_thr * mean,
        the fp32 product of bfs_cluster.cpp:73-79"""
        g = self.grouping_cfg
        npoint_thr = _cfg(g, 'npoint_thr')
        class_mean = torch.tensor(_cfg(g, 'class_numpoint_mean'), dtype=torch.float32)
        assert class_mean.size(0) == self.semantic_classes
        ignore = set(_cfg(g, 'ignore_classes'))
        classes = [c for c in range(self.semantic_classes) if c not in ignore]
        const = self.__dict__.setdefault('_grouping_const', {})
        ck = (str(dev), tuple(classes), float(npoint_thr), tuple(class_mean.tolist()))
        if ck not in const:
            m = class_mean[classes].numpy()
            thr = np.where(m == np.float32(-1), np.float32(npoint_thr), np.float32(npoint_thr) * m)
            const[ck] = (torch.tensor(classes, device=dev),
                         torch.from_numpy(thr.astype(np.float32)).to(dev),
                         torch.zeros(2, dtype=torch.int32, device=dev),
                         torch.tensor(classes, dtype=torch.int32, device=dev))
            torch.cuda.current_stream().synchronize()      # once: other streams may use them next
        return const[ck]

    def _native_proposals(self, semantic_scores, pt_offsets, batch_idxs, coords_float, batch_size):
        """forward_grouping's result from the native driver stopped after the clustering (the training
        step voxelises the proposals itself, with rand_quantize): same values, one C call"""
        from . import native_scan as NS
        g = self.grouping_cfg
        dev = semantic_scores.device
        _, seg_thr, _, cls32 = self._grouping_constants(dev)
        scores = semantic_scores.float().softmax(dim=-1)
        cfg = NS.GroupingCfg(
            n_points=scores.size(0), n_sem_classes=scores.size(1), n_seg=cls32.numel(),
            seg_class=cls32.data_ptr(), seg_thr=seg_thr.data_ptr(), score_thr=_cfg(g, 'score_thr'),
            min_npoint=_cfg(self.test_cfg, 'min_npoint'), radius=_cfg(g, 'radius'), batch_size=int(batch_size),
            voxel_scale=1.0, voxel_shape=0, feat_channels=1)
        r = NS.grouping(cfg, scores, pt_offsets.float().contiguous(), coords_float.contiguous(),
                        batch_idxs.int().contiguous(), None)
        if r is None:
            return (torch.zeros((0, 2), dtype=torch.int32, device=dev),
                    torch.zeros((0, ), dtype=torch.int32, device=dev))
        return r['proposals_idx'], r['proposals_offset']

    def _native_grouping_and_refinement(self, semantic_scores, pt_offsets, batch_idxs, coords_float,
                                        output_feats, batch_size):
        """forward_grouping + clusters_voxelization + forward_instance with the grouping head and the
        proposal voxelisation in one C call (native_scan.grouping); the dense heads stay torch
        modules.  -> (proposals_idx, cls_scores, iou_scores, mask_scores) or None when there is no
        proposal (the caller's module path builds the reference's dummy tensor)."""
        from . import native_scan as NS
        g, v = self.grouping_cfg, self.instance_voxel_cfg
        dev = semantic_scores.device
        _, seg_thr, _, cls32 = self._grouping_constants(dev)
        scores = semantic_scores.float().softmax(dim=-1)
        offs = pt_offsets.float().contiguous()
        feats = output_feats.contiguous()
        cfg = NS.GroupingCfg(
            n_points=scores.size(0), n_sem_classes=scores.size(1), n_seg=cls32.numel(),
            seg_class=cls32.data_ptr(), seg_thr=seg_thr.data_ptr(), score_thr=_cfg(g, 'score_thr'),
            min_npoint=_cfg(self.test_cfg, 'min_npoint'), radius=_cfg(g, 'radius'), batch_size=int(batch_size),
            voxel_scale=_cfg(v, 'scale'), voxel_shape=_cfg(v, 'spatial_shape'), feat_channels=feats.size(1))
        if _cfg(g, 'with_pyramid', False) or _cfg(g, 'with_octree', False):
            # SoftGroup++: the per-class loop of _grouping_per_class (reference softgroup.py:443-463) inside C
            cfg = NS.GroupingPPCfg(base=cfg, with_pyramid=int(bool(_cfg(g, 'with_pyramid', False))),
                                   with_octree=int(bool(_cfg(g, 'with_octree', False))), lvl_fusion=0,
                                   radius=float(_cfg(g, 'radius')),
                                   base_size=float(_cfg(g, 'pyramid_base_size', 0.02)))
        r = NS.grouping(cfg, scores, offs, coords_float.contiguous(), batch_idxs.int().contiguous(), feats)
        if r is None:
            return None
        ss = _cfg(v, 'spatial_shape')
        x = spconv.SparseConvTensor(r['voxel_feats'], r['voxel_coords'], [ss] * 3, r['n_proposals'])
        ex = self.__dict__.get('_tiny_exec')
        if ex is None:
            ex = self.__dict__['_tiny_exec'] = UNetExecutor(self.tiny_unet, None,
                                                             self.tiny_unet_outputlayer)
        if ex.usable(x.features):
            vfeats = ex(x)
        else:
            vfeats = self.tiny_unet_outputlayer(self.tiny_unet(x)).features
        mask_scores = self.mask_linear(vfeats)[r['point_to_voxel'].long()]
        pooled = ops.global_avg_pool(vfeats, r['voxel_offsets'])
        return r['proposals_idx'], self.cls_linear(pooled), self.iou_score_linear(pooled), mask_scores

    def _unet_features(self, x):
        """input_conv -> unet -> output_layer.  Inference runs in the native executor (one C call,
        same kernels); anything it does not cover (training, autocast dtypes) takes the modules."""
        ex = self.__dict__.get('_backbone_exec')
        if ex is None:
            ex = self.__dict__['_backbone_exec'] = UNetExecutor(self.unet, self.input_conv,
                                                                 self.output_layer)
        if self.use_executor and ex.usable(x.features):
            return ex(x)
        if self.use_train_executor and self._train_exec('_backbone_train_exec', self.unet, self.input_conv,
                                                        self.output_layer).usable(x.features):
            return self.__dict__['_backbone_train_exec'](x)      # backbone not frozen, train() mode
        return self.output_layer(self.unet(self.input_conv(x))).features

    def _train_exec(self, slot, unet, input_conv, output_layer):
        ex = self.__dict__.get(slot)
        if ex is None:
            ex = self.__dict__[slot] = UNetTrainExecutor(unet, input_conv, output_layer)
        return ex

    def forward_backbone(self, input, input_map, x4_split=False, lvl_fusion=False):
        return self._forward_backbone(input, input_map, x4_split, lvl_fusion)[:3]

    def _forward_backbone(self, input, input_map, x4_split=False, lvl_fusion=False):
        """-> (semantic_scores, pt_offsets, output_feats, semantic_preds or None).  In inference the
        devoxelize gather, both point-wise heads and the arg-max run as ONE kernel (sg_pointwise_heads);
        training, half precision and anything but the reference's MLP structure take the modules."""
        if x4_split:
            assert not lvl_fusion, 'x4_split not support lvl_fusion'
            output_feats = self.merge_4_parts(self.forward_4_parts(input, input_map))
        else:
            output_feats = self._unet_features(input)
            heads = self._fused_heads(output_feats)
            if heads is not None:
                return self._run_fused_heads(heads, output_feats, None if lvl_fusion else input_map)
            if not lvl_fusion:
                output_feats = _take_rows(output_feats, input_map)       # devoxelize
        semantic_scores = self.semantic_linear(output_feats)
        pt_offsets = self.offset_linear(output_feats)
        return semantic_scores, pt_offsets, output_feats, None

    # ---- point-wise heads as one kernel (csrc/heads.hip)
    def _fused_heads(self, feats):
        """the two MLP heads as C descriptors, or None when the fused kernel does not apply"""
        if not (self.use_fused_heads and feats.is_cuda and feats.dtype == torch.float32
                and not torch.is_grad_enabled() and feats.shape[1] in (16, 32)):
            return None
        from ..spconv import core as spcore
        out = []
        for mlp in (self.semantic_linear, self.offset_linear):
            mods = list(mlp._modules.values())
            if not (len(mods) == 4 and isinstance(mods[0], nn.Linear) and isinstance(mods[1], nn.BatchNorm1d)
                    and isinstance(mods[2], nn.ReLU) and isinstance(mods[3], nn.Linear)):
                return None
            l1, bn, _, l2 = mods
            if (bn.training or bn.running_mean is None or l1.bias is None or l2.bias is None
                    or l1.weight.shape != (feats.shape[1], feats.shape[1]) or l2.weight.shape[1] != feats.shape[1]):
                return None
            ts = (l1.weight, l1.bias, l2.weight, l2.bias)
            if not all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in ts):
                return None
            scale, shift = spcore._bn_affine(bn)
            out.append((l1.weight, l1.bias, scale, shift, l2.weight, l2.bias))
        if not (out[0][4].shape[0] <= 32 and out[1][4].shape[0] <= 4):
            return None
        return out

    def _run_fused_heads(self, heads, voxel_feats, v2p_map):
        from .. import _lib as L
        import ctypes as C
        voxel_feats = voxel_feats.contiguous()
        dev = voxel_feats.device
        c = voxel_feats.shape[1]
        if v2p_map is not None:
            v2p_map = v2p_map.contiguous()
            if v2p_map.dtype not in (torch.int32, torch.int64):
                v2p_map = v2p_map.long()
        n = voxel_feats.shape[0] if v2p_map is None else v2p_map.numel()
        descs = []
        for w1, b1, sc, sh, w2, b2 in heads:
            d = _Mlp2()
            d.w1, d.b1, d.bn_scale, d.bn_shift, d.w2, d.b2 = (t.data_ptr() for t in (w1, b1, sc, sh, w2, b2))
            d.out = w2.shape[0]
            descs.append(d)
        sem = torch.empty((n, descs[0].out), dtype=torch.float32, device=dev)
        off = torch.empty((n, descs[1].out), dtype=torch.float32, device=dev)
        preds = torch.empty((n, ), dtype=torch.int64, device=dev)
        out_feats = voxel_feats if v2p_map is None else torch.empty((n, c), dtype=torch.float32, device=dev)
        L.check(L.lib().sg_pointwise_heads(
            L.ptr(voxel_feats), L.ptr(v2p_map), int(v2p_map is not None and v2p_map.dtype == torch.int64), n, c,
            C.byref(descs[0]), C.byref(descs[1]), None if v2p_map is None else L.ptr(out_feats), L.ptr(sem),
            L.ptr(off), L.ptr(preds), L.stream()), 'sg_pointwise_heads')
        return sem, off, out_feats, preds

    def forward_4_parts(self, x, input_map):
        """S3DIS: the scene arrives as 4 interleaved sub-clouds (batch ids 0..3); run them one at a
        time with batch id 0 and stack the voxel features (reference softgroup.py:380-395)."""
        outs = []
        for part in range(4):
            sel = x.indices[:, 0] == part
            coords = x.indices[sel].clone()
            coords[:, 0] = 0
            sub = spconv.SparseConvTensor(x.features[sel], coords, x.spatial_shape, 1)
            outs.append(self._unet_features(sub))
        return _take_rows(torch.cat(outs, dim=0), input_map)

    def merge_4_parts(self, x):
        """inverse of the stride-4 interleave of data/s3dis.py:50-54"""
        n = x.size(0)
        sizes = [(n - p + 3) // 4 for p in range(4)]
        out = torch.zeros_like(x)
        for p, chunk in enumerate(torch.split(x, sizes)):
            out[p::4] = chunk
        return out

    # ------------------------------------------------------------------ grouping
    @force_fp32(apply_to=('semantic_scores', 'pt_offsets'))
    def forward_grouping(self, semantic_scores, pt_offsets, batch_idxs, coords_float,
                         grouping_cfg=None, lvl_fusion=False, batch_size=None):
        """-> proposals_idx int32 [S,2] (proposal id, point idx), proposals_offset int32 [nP+1].
        Same values and order as the reference (softgroup.py:411-480), kept on the GPU."""
        g = self.grouping_cfg
        dev = semantic_scores.device
        radius, mean_active = _cfg(g, 'radius'), _cfg(g, 'mean_active')
        npoint_thr = _cfg(g, 'npoint_thr')
        with_pyramid = _cfg(g, 'with_pyramid', False)
        with_octree = _cfg(g, 'with_octree', False)
        base_size = _cfg(g, 'pyramid_base_size', 0.02)
        class_mean = torch.tensor(_cfg(g, 'class_numpoint_mean'), dtype=torch.float32)
        assert class_mean.size(0) == self.semantic_classes
        ignore = set(_cfg(g, 'ignore_classes'))
        classes = [c for c in range(self.semantic_classes) if c not in ignore]
        min_npoint = _cfg(self.test_cfg, 'min_npoint')
        if batch_size is None:       # (the reference reads it back from the GPU every time)
            batch_size = int(batch_idxs.max()) + 1
        scores = semantic_scores.softmax(dim=-1)

        if with_pyramid or with_octree:
            return self._grouping_per_class(scores, pt_offsets, batch_idxs, coords_float, classes,
                                            class_mean, batch_size, radius, mean_active, npoint_thr,
                                            with_pyramid, with_octree, base_size, min_npoint,
                                            lvl_fusion)

        # ---- all classes at once: segment s = position of the class in `classes`
        # (constants of the configuration live on the device once, not re-uploaded per scan)
        cls_t, seg_thr, dummy_offsets, _ = self._grouping_constants(dev)
        sel = scores[:, cls_t].t() > _cfg(g, 'score_thr')                  # [n_seg, N]
        sel &= (sel.sum(1, keepdim=True, dtype=torch.int32) >= min_npoint)   # small classes are skipped
        seg, obj = sel.nonzero(as_tuple=True)                              # class-major, point-ascending
        if obj.numel() == 0:
            return (torch.zeros((0, 2), dtype=torch.int32, device=dev),
                    torch.zeros((0, ), dtype=torch.int32, device=dev))
        pts = (coords_float[obj] + pt_offsets[obj]).contiguous()
        seg32 = seg.int()
        key = (seg32 * batch_size + batch_idxs[obj].int()).contiguous()    # never mix classes/scenes
        nbr_idx, start_len = ops.ballquery_batch_p(pts, key, dummy_offsets, radius, mean_active)
        proposals_idx, proposals_offset = ops.bfs_cluster_segments(
            nbr_idx, start_len, seg_thr, seg32.contiguous(), ops.LISTS_SORTED | ops.LISTS_RADIUS)
        if proposals_idx.shape[0] == 0:
            return proposals_idx, torch.zeros((0, ), dtype=torch.int32, device=dev)
        proposals_idx[:, 1] = obj[proposals_idx[:, 1].long()].int()       # local -> scene point index
        return proposals_idx, proposals_offset

    def _grouping_per_class(self, scores, pt_offsets, batch_idxs, coords_float, classes, class_mean,
                            batch_size, radius0, mean_active, npoint_thr, with_pyramid, with_octree,
                            base_size, min_npoint, lvl_fusion):
        """SoftGroup++ path (octree query / pyramid levels depend on the per-class point count,
        reference softgroup.py:443-463): one class at a time, everything on the GPU."""
        dev = scores.device
        idx_list, off_list = [], []
        n_prop, n_pts = 0, 0
        # the selected points of ALL classes by one nonzero (class-major, points ascending = what the
        # reference's per-class `.nonzero()` yields) and ONE read-back of the per-class counts; the
        # per-class loop then slices (it used to issue a compare + nonzero + host sync per class: 15
        # syncs per STPLS3D scan before any class was known to be large enough)
        cls_t = torch.tensor(classes, device=dev)
        sel = scores[:, cls_t].t() > _cfg(self.grouping_cfg, 'score_thr')          # [n_classes, N]
        counts = sel.sum(1, dtype=torch.int32).tolist()
        _, obj_all = sel.nonzero(as_tuple=True)
        starts = np.concatenate([[0], np.cumsum(counts)])
        for ci, class_id in enumerate(classes):
            if counts[ci] < min_npoint:
                continue
            obj = obj_all[int(starts[ci]):int(starts[ci + 1])]
            b_, c_, o_ = batch_idxs[obj], coords_float[obj], pt_offsets[obj]
            radius, level, l2p_map = radius0, 1, None
            if with_pyramid:
                level = self.get_level(c_.size(0))
                radius = radius0 * level
                if level > 1 or not lvl_fusion:
                    c_, o_, b_, l2p_map = self.pyramid_map(c_, o_, b_, level, base_size, batch_size=batch_size)
            if batch_size == 1:          # (no bincount / cumsum for a single scene)
                offs = torch.tensor([0, b_.size(0)], dtype=torch.int32, device=dev)
            else:
                offs = self.get_batch_offsets(b_, batch_size)
            nbr, start_len = ops.ball_query((c_ + o_).contiguous(), b_.int().contiguous(), offs,
                                            radius, mean_active, with_octree=with_octree)
            pidx, poff = ops.bfs_cluster(class_mean, nbr, start_len, npoint_thr, class_id)
            if l2p_map is not None:
                pidx, poff = self.pyramid_inverse_map(pidx, poff, c_.size(0), l2p_map)
            if pidx.size(0) == 0:
                continue
            pidx = pidx.clone()
            pidx[:, 1] = obj[pidx[:, 1].long()].int()
            pidx[:, 0] += n_prop
            idx_list.append(pidx)
            off_list.append(poff[1:] + n_pts if off_list else poff + n_pts)
            n_prop += poff.numel() - 1
            n_pts += pidx.size(0)
        if not idx_list:
            return (torch.zeros((0, 2), dtype=torch.int32, device=dev),
                    torch.zeros((0, ), dtype=torch.int32, device=dev))
        return torch.cat(idx_list, 0), torch.cat(off_list).int()

    def get_level(self, num_points):
        if num_points > 1000000:
            return 3
        return 2 if num_points > 100000 else 1

    def pyramid_map(self, coords_float, pt_offsets, batch_idxs, level=1, base_size=0.02, batch_size=None):
        """coarse voxels for big classes; .long() truncates toward zero like the reference
        (softgroup.py:491-498).  ``batch_size``: what the reference reads back from
        the GPU (`batch_idxs[-1] + 1`) when the caller knows it -- the voxel index depends on it only
        through the key range"""
        vox = (coords_float / (base_size * level)).long()
        vox = torch.cat([batch_idxs[:, None].long(), vox], dim=1).contiguous()
        if batch_size is None:
            batch_size = int(batch_idxs[-1].item()) + 1
        out_coords, l2p_map, p2l_map = ops.voxelization_idx(vox, int(batch_size))
        coords_float = ops.voxelization(coords_float.contiguous(), p2l_map)
        pt_offsets = ops.voxelization(pt_offsets.contiguous(), p2l_map)
        return coords_float, pt_offsets, out_coords[:, 0].int(), l2p_map

    def pyramid_inverse_map(self, proposals_idx, proposals_offset, num_points, l2p_map):
        """expand level voxels back to the class's points: proposal p contains point j iff it
        contains voxel l2p_map[j]; rows ordered (proposal, point) ascending like the reference's
        dense nonzero (softgroup.py:500-507).  The reference builds an int [nProposal, n] matrix
        for this (GBs at STPLS3D scale); clusters of one class are disjoint, so a voxel -> proposal
        table and one stable sort of the points by proposal give the same rows in O(n) memory."""
        dev = proposals_idx.device
        n_prop = proposals_offset.numel() - 1
        if proposals_idx.is_cuda and l2p_map.is_cuda:
            # one C call (table, gather, stable sort by proposal, counts): csrc/octree.hip
            return ops.pyramid_inverse_map(proposals_idx, n_prop, l2p_map, num_points)
        prop_of_voxel = torch.full((num_points, ), -1, dtype=torch.long, device=dev)
        prop_of_voxel[proposals_idx[:, 1].long()] = proposals_idx[:, 0].long()
        prop_of_point = prop_of_voxel[l2p_map.long().to(dev)]              # [n_orig]
        pts = (prop_of_point >= 0).nonzero().view(-1)                       # ascending point index
        order = torch.argsort(prop_of_point[pts], stable=True)              # proposal-major, points ascending
        pts = pts[order]
        prop = prop_of_point[pts]
        pidx = torch.stack([prop, pts], 1).int()
        counts = torch.bincount(prop, minlength=n_prop)
        poff = torch.cat([counts.new_zeros(1), torch.cumsum(counts, 0)]).int()
        return pidx, poff

    def get_batch_offsets(self, batch_idxs, bs):
        counts = torch.bincount(batch_idxs.long(), minlength=bs)
        offs = torch.zeros(bs + 1, dtype=torch.int32, device=batch_idxs.device)
        offs[1:] = torch.cumsum(counts, 0)
        return offs

    # ------------------------------------------------------------------ proposal voxelisation
    @force_fp32(apply_to='feats')
    def clusters_voxelization(self, clusters_idx, clusters_offset, feats, coords, scale,
                              spatial_shape, rand_quantize=False):
        dev = feats.device
        if clusters_idx.size(0) == 0:        # dummy 2-voxel tensor (reference softgroup.py:664-673)
            far = spatial_shape - 1
            dummy = torch.tensor([[0, 0, 0, 0], [0, far, far, far]], dtype=torch.int32, device=dev)
            t = spconv.SparseConvTensor(feats[0:2], dummy, [spatial_shape] * 3, 1)
            return t, feats.new_zeros((1, ), dtype=torch.long)

        clusters_idx = clusters_idx.to(dev)
        clusters_offset = clusters_offset.to(dev).int().contiguous()
        cluster_of = clusters_idx[:, 0].long()
        pt = clusters_idx[:, 1].contiguous()
        feats = _take_rows(feats, pt)
        coords = _take_rows(coords, pt)

        lo = ops.sec_min(coords, clusters_offset)
        hi = ops.sec_max(coords, clusters_offset)
        # 0.01 keeps voxel coords < spatial_shape
        cscale = 1 / ((hi - lo) / spatial_shape).max(1)[0] - 0.01
        cscale = torch.clamp(cscale, min=None, max=scale)
        lo = lo * cscale[:, None]
        hi = hi * cscale[:, None]
        coords = coords * cscale[cluster_of][:, None]
        if rand_quantize:
            # two draws from the CPU generator, like the reference (`torch.rand(3).cuda()`,
            # softgroup.py:692-693): the same torch.manual_seed gives the same crops
            span = hi - lo
            lo -= torch.clamp(spatial_shape - span - 0.001, min=0) * torch.rand(3).to(dev)
            lo -= torch.clamp(spatial_shape - span + 0.001, max=0) * torch.rand(3).to(dev)
        coords -= lo[cluster_of]
        assert coords.shape.numel() == ((coords >= 0) * (coords < spatial_shape)).sum()
        vox = torch.cat([cluster_of.view(-1, 1), coords.long()], 1).contiguous()

        n_prop = int(clusters_offset.numel()) - 1          # == int(clusters_idx[-1,0]) + 1
        out_coords, inp_map, out_map = ops.voxelization_idx(vox, n_prop)
        out_feats = ops.voxelization(feats, out_map)
        t = spconv.SparseConvTensor(out_feats, out_coords.int(), [spatial_shape] * 3, n_prop)
        return t, inp_map

    @force_fp32(apply_to=('x'))
    def global_pool(self, x, expand=False):
        ids = x.indices[:, 0]
        # voxels per proposal; scatter_add into a fixed-size tensor (bincount would read the maximum
        # back to the host first)
        counts = torch.zeros(x.batch_size, dtype=torch.long, device=ids.device).scatter_add_(
            0, ids.long(), torch.ones_like(ids, dtype=torch.long))
        offsets = torch.cat([counts.new_zeros(1), torch.cumsum(counts, 0)]).int()
        pooled = ops.global_avg_pool(x.features.contiguous(), offsets)
        if not expand:
            return pooled
        x.features = torch.cat((x.features, pooled[ids.long()]), dim=1)
        return x

    def forward_instance(self, inst_feats, inst_map):
        ex = self.__dict__.get('_tiny_exec')
        if ex is None:
            ex = self.__dict__['_tiny_exec'] = UNetExecutor(self.tiny_unet, None,
                                                             self.tiny_unet_outputlayer)
        if self.use_executor and ex.usable(inst_feats.features):
            feats = inst_feats.replace_feature(ex(inst_feats))
        elif self.use_train_executor and self._train_exec('_tiny_train_exec', self.tiny_unet, None,
                                                          self.tiny_unet_outputlayer).usable(inst_feats.features):
            # train() mode: forward and backward of the tiny U-Net are one C call each
            feats = inst_feats.replace_feature(self.__dict__['_tiny_train_exec'](inst_feats))
        else:
            feats = self.tiny_unet_outputlayer(self.tiny_unet(inst_feats))
        inst_map = inst_map.long()
        mask_scores = self.mask_linear(feats.features)[inst_map]
        instance_batch_idxs = feats.indices[:, 0][inst_map]
        pooled = self.global_pool(feats)
        return instance_batch_idxs, self.cls_linear(pooled), self.iou_score_linear(pooled), mask_scores

    # ------------------------------------------------------------------ results
    @force_fp32(apply_to=('semantic_preds', 'offset_preds'))
    def get_point_wise_results(self, coords_float, color_feats, semantic_preds, offset_preds,
                               offset_labels, v2p_map, lvl_fusion, _device=False):
        if lvl_fusion:
            semantic_preds = semantic_preds[v2p_map.long()]
            offset_preds = offset_preds[v2p_map.long()]
        res = dict(coords_float=coords_float, color_feats=color_feats, semantic_preds=semantic_preds,
                   offset_preds=offset_preds, offset_labels=offset_labels)
        return res if _device else to_host(res)

    @force_fp32(apply_to=('semantic_scores', 'cls_scores', 'iou_scores', 'mask_scores'))
    def get_instances(self, scan_id, proposals_idx, semantic_scores, cls_scores, iou_scores,
                      mask_scores, v2p_map=None, lvl_fusion=False, _panoptic_sem=None):
        """Same instances, order and RLE strings as the reference (softgroup.py:537-604).  For
        instance class i a proposal survives iff cls_score > cls_score_thr and its mask
        (mask_score > mask_score_thr) has >= min_npoint points; masks are encoded from sorted
        (proposal, point) pairs -- no dense [nProposal, N] matrix is built."""
        if proposals_idx.size(0) == 0:
            return []
        tcfg = self.test_cfg
        dev = cls_scores.device
        n_inst, n_pts = cls_scores.size(0), semantic_scores.size(0)
        n_out = v2p_map.numel() if lvl_fusion else n_pts
        cls_prob = cls_scores.softmax(1)
        prop, pt = proposals_idx[:, 0].long().to(dev), proposals_idx[:, 1].long().to(dev)
        if not lvl_fusion and not self.sem2ins_classes and cls_scores.is_cuda and self.use_native_scan:
            # one C call (csrc/scan_exec.hip): per-(proposal, class) point counts, the keep table and
            # its class-major numbering on the device, one bit row per kept instance -> runs -> RLE
            # text; labels, scores, offsets and the text arrive in one pinned host buffer
            from . import native_scan as NS
            nc = self.instance_classes
            pairs = proposals_idx.int().contiguous()
            ms = mask_scores.float().contiguous()
            cfg = NS.InstancesCfg(
                n_proposals=n_inst, n_classes=nc, score_stride=ms.size(1), sum_npoint=pairs.size(0),
                n_points=n_out, cls_score_thr=_cfg(tcfg, 'cls_score_thr'),
                mask_score_thr=_cfg(tcfg, 'mask_score_thr'), min_npoint=_cfg(tcfg, 'min_npoint'))
            assert cls_prob.size(1) == ms.size(1) == iou_scores.size(1)
            pan = None
            if _panoptic_sem is not None:       # fuse on the device while the bit rows are there
                pan = dict(semantic_preds=_panoptic_sem,
                           cls_offset=self.semantic_classes - self.instance_classes - 1,
                           skip_iou=_cfg(tcfg, 'panoptic_skip_iou'), semantic_classes=self.semantic_classes)
            label, conf, text, off, fused = NS.instances(cfg, pairs, ms, cls_prob.float().contiguous(),
                                                         iou_scores.float().contiguous(), pan)
            label = label.astype(np.int64)
            insts = [dict(scan_id=scan_id, label_id=label[k], conf=conf[k],
                          pred_mask=dict(length=int(n_out), counts=text[off[k]:max(off[k + 1] - 1, off[k])]))
                     for k in range(label.shape[0])]
            if _panoptic_sem is not None:
                return insts, fused
            return insts
        if not lvl_fusion and not self.sem2ins_classes and cls_scores.is_cuda:
            # all instance classes in one pass on the GPU (csrc/instances.hip): per-(proposal,
            # class) point counts, then one bit row per KEPT instance -> runs; two small read-backs
            from .. import _lib as L
            lib = L.lib()
            nc = self.instance_classes
            S = proposals_idx.size(0)
            pairs = proposals_idx.int().contiguous()
            ms = mask_scores.contiguous()
            mthr = float(_cfg(tcfg, 'mask_score_thr'))
            npoint = torch.empty((n_inst, nc), dtype=torch.int32, device=dev)
            L.check(lib.sg_instance_npoint(L.ptr(pairs), L.ptr(ms), S, ms.size(1), nc, mthr, n_inst,
                                           L.ptr(npoint), L.stream()), 'sg_instance_npoint')
            # one read-back of the three small [n_inst, nc] tables; which (class, proposal) pairs are
            # kept, their order (class-major, the reference's) and the run capacity are host work
            probs = cls_prob[:, :nc].contiguous()
            score = probs * iou_scores[:, :nc].clamp(0, 1)
            host = torch.stack([npoint.view(torch.float32), probs, score]).cpu().numpy()
            npoint_h = host[0].view(np.int32)
            keep = (host[1] > _cfg(tcfg, 'cls_score_thr')) & (npoint_h >= _cfg(tcfg, 'min_npoint'))
            kept = np.argwhere(keep.T)                      # rows (class, proposal), class-major
            n_kept = kept.shape[0]
            if n_kept == 0:
                return []
            inst_of_h = np.full((nc, n_inst), -1, dtype=np.int32)
            inst_of_h[kept[:, 0], kept[:, 1]] = np.arange(n_kept, dtype=np.int32)
            inst_of = torch.from_numpy(inst_of_h).to(dev, non_blocking=True)
            cap = int(npoint_h[keep].sum())                 # every kept point is at most one run
            starts = torch.empty(max(cap, 1), dtype=torch.int32, device=dev)
            ends = torch.empty(max(cap, 1), dtype=torch.int32, device=dev)
            bounds = torch.empty(n_kept + 1, dtype=torch.int64, device=dev)
            ws = L.workspace(lib.sg_instance_runs_workspace_bytes(n_kept, n_out), dev)
            L.check(lib.sg_instance_runs(L.ptr(pairs), L.ptr(ms), S, ms.size(1), nc, mthr,
                                         L.ptr(inst_of), n_inst, n_kept, n_out, L.ptr(starts),
                                         L.ptr(ends), L.ptr(bounds), cap, L.ptr(ws), ws.numel(),
                                         L.stream()), 'sg_instance_runs')
            # the "start len ..." text is written on the device too: what travels to the host is the
            # text itself and one offset per instance
            tcap = int(lib.sg_rle_format_device_text_bytes(cap, n_out))
            text = torch.empty(tcap, dtype=torch.uint8, device=dev)
            text_off = torch.empty(n_kept + 1, dtype=torch.int64, device=dev)
            ws = L.workspace(lib.sg_rle_format_device_workspace_bytes(cap), dev)
            L.check(lib.sg_rle_format_device(L.ptr(starts), L.ptr(ends), L.ptr(bounds), n_kept, cap,
                                             n_out, L.ptr(text), tcap, L.ptr(text_off), L.ptr(ws),
                                             ws.numel(), L.stream()), 'sg_rle_format_device')
            cls_pred = kept[:, 0] + 1
            score_pred = host[2][kept[:, 1], kept[:, 0]]
            o = text_off.cpu().tolist()
            masks = rle_text_to_dicts(n_out, text, o)
            return [dict(scan_id=scan_id, label_id=cls_pred[k], conf=score_pred[k],
                         pred_mask=masks[k]) for k in range(n_kept)]
        sem_pred = semantic_scores.max(1)[1]
        if lvl_fusion:
            # every voxel stands for the points mapped to it: expand pairs voxel -> points
            order = torch.argsort(v2p_map.long(), stable=True)
            vcount = torch.bincount(v2p_map.long(), minlength=n_pts)
            vstart = torch.cumsum(vcount, 0) - vcount
        cls_all, score_all, runs = [], [], []
        for i in range(self.instance_classes):
            if i in self.sem2ins_classes:
                m = (sem_pred == i)
                if lvl_fusion:
                    m = m[v2p_map.long()]
                cls_all.append(torch.tensor([i + 1], dtype=torch.long))
                score_all.append(torch.tensor([1.], dtype=torch.float32))
                runs.append(_runs_of_pairs(torch.zeros_like(m.nonzero().view(-1)),
                                           m.nonzero().view(-1), 1))
                continue
            score = cls_prob[:, i] * iou_scores[:, i].clamp(0, 1)
            on = mask_scores[:, i] > _cfg(tcfg, 'mask_score_thr')
            p_on, q_on = prop[on], pt[on]
            if lvl_fusion:
                rep = vcount[q_on]
                p_on = torch.repeat_interleave(p_on, rep)
                base = torch.repeat_interleave(vstart[q_on], rep)
                within = torch.arange(p_on.numel(), device=dev) - torch.repeat_interleave(
                    torch.cumsum(rep, 0) - rep, rep)
                q_on = order[base + within]
            # a (proposal, point) pair can repeat only if the proposal lists a point twice: it cannot
            npoint = torch.bincount(p_on, minlength=n_inst)
            keep = (cls_prob[:, i] > _cfg(tcfg, 'cls_score_thr')) & (npoint >= _cfg(tcfg, 'min_npoint'))
            kept = keep.nonzero().view(-1)
            new_id = torch.full((n_inst, ), -1, dtype=torch.long, device=dev)
            new_id[kept] = torch.arange(kept.numel(), device=dev)
            sel = new_id[p_on] >= 0
            cls_all.append(torch.full((kept.numel(), ), i + 1, dtype=torch.long))
            score_all.append(score[kept].cpu())
            runs.append(_runs_of_pairs(new_id[p_on[sel]], q_on[sel], kept.numel()))
        cls_pred = torch.cat(cls_all).numpy()
        score_pred = torch.cat(score_all).numpy()
        instances, k = [], 0
        for starts, lens, bounds in runs:
            for j in range(len(bounds) - 1):
                a, b = bounds[j], bounds[j + 1]
                instances.append(dict(scan_id=scan_id, label_id=cls_pred[k], conf=score_pred[k],
                                      pred_mask=rle_encode_runs(n_out, starts[a:b], lens[a:b])))
                k += 1
        return instances

    def panoptic_fusion(self, semantic_preds, instance_preds):
        """paste instances by descending confidence (reference softgroup.py:606-639)"""
        cls_offset = self.semantic_classes - self.instance_classes - 1
        panoptic_cls = semantic_preds.copy().astype(np.uint32)
        panoptic_ids = np.zeros_like(semantic_preds).astype(np.uint32)
        taken = np.zeros_like(semantic_preds, dtype=bool)
        next_id = 1
        for i in np.argsort([x['conf'] for x in instance_preds])[::-1]:
            inst = instance_preds[i]
            # the mask as the list of its points (from the runs): the reference decodes every mask
            # to a dense N-vector, O(N) per instance; the arithmetic below is the same
            tok = np.array(inst['pred_mask']['counts'].split(), dtype=np.int64)
            starts, lens = tok[0::2] - 1, tok[1::2]
            pts = np.repeat(starts - np.cumsum(lens) + lens, lens) + np.arange(int(lens.sum()))
            hit = taken[pts]
            overlap = hit.sum()
            if overlap / (pts.size + 1e-5) > _cfg(self.test_cfg, 'panoptic_skip_iou'):
                continue
            paste = pts[~hit]
            panoptic_cls[paste] = inst['label_id'] + cls_offset
            panoptic_ids[paste] = next_id
            taken[paste] = 1
            next_id += 1
        ignore = (panoptic_cls >= 11) & (panoptic_ids == 0)     # thing classes without an id
        out = (panoptic_cls & 0xFFFF) | (panoptic_ids << 16)
        out[ignore] = self.semantic_classes
        return out.astype(np.uint32)

    def get_gt_instances(self, semantic_labels, instance_labels, _device=False):
        """ScanNet encoding sem*1000 + inst, 0 = ignore (reference softgroup.py:641-653)"""
        shift = self.semantic_classes - self.instance_classes
        sem = semantic_labels - shift + 1
        sem[sem < 0] = 0
        instance_labels = instance_labels + 1      # (the reference increments its private copy in place)
        gt = sem * 1000 + instance_labels
        gt[instance_labels < 0] = 0
        return gt if _device else gt.cpu().numpy()

    # ------------------------------------------------------------------ training
    @cuda_cast
    def forward_train(self, batch_idxs, voxel_coords, p2v_map, v2p_map, coords_float, feats,
                      semantic_labels, instance_labels, instance_pointnum, instance_cls,
                      pt_offset_labels, spatial_shape, batch_size, **kwargs):
        losses = {}
        if self.with_coords:
            feats = torch.cat((feats, coords_float), 1)
        voxel_feats = ops.voxelization(feats, p2v_map)
        x = spconv.SparseConvTensor(voxel_feats, voxel_coords.int(), spatial_shape, batch_size)
        semantic_scores, pt_offsets, output_feats = self.forward_backbone(x, v2p_map)
        losses.update(self.point_wise_loss(semantic_scores, pt_offsets, semantic_labels,
                                           instance_labels, pt_offset_labels))
        if not self.semantic_only:
            with torch.no_grad():
                g = self.grouping_cfg
                n_seg = self.semantic_classes - len(set(_cfg(g, 'ignore_classes')))
                # (same bound as _native_scan_usable: the driver holds at most 32 grouped classes)
                if (self.use_native_scan and semantic_scores.is_cuda and not _cfg(g, 'with_pyramid', False)
                        and not _cfg(g, 'with_octree', False) and 0 < n_seg <= 32):
                    proposals_idx, proposals_offset = self._native_proposals(
                        semantic_scores.detach(), pt_offsets.detach(), batch_idxs, coords_float, batch_size)
                else:
                    proposals_idx, proposals_offset = self.forward_grouping(
                        semantic_scores.detach(), pt_offsets.detach(), batch_idxs, coords_float,
                        self.grouping_cfg, batch_size=batch_size)
            max_prop = _cfg(self.train_cfg, 'max_proposal_num')
            if proposals_offset.shape[0] > max_prop:
                proposals_offset = proposals_offset[:max_prop + 1]
                proposals_idx = proposals_idx[:int(proposals_offset[-1])]
                assert proposals_idx.shape[0] == proposals_offset[-1]
            inst_feats, inst_map = self.clusters_voxelization(
                proposals_idx, proposals_offset, output_feats, coords_float, rand_quantize=True,
                **self.instance_voxel_cfg)
            instance_batch_idxs, cls_scores, iou_scores, mask_scores = self.forward_instance(
                inst_feats, inst_map)
            losses.update(self.instance_loss(cls_scores, mask_scores, iou_scores, proposals_idx,
                                             proposals_offset, instance_labels, instance_pointnum,
                                             instance_cls, instance_batch_idxs))
        return self.parse_losses(losses)

    def point_wise_loss(self, semantic_scores, pt_offsets, semantic_labels, instance_labels,
                        pt_offset_labels):
        weight = None
        if self.semantic_weight:
            weight = torch.tensor(self.semantic_weight, dtype=torch.float, device=semantic_scores.device)
        losses = dict(semantic_loss=_cross_entropy(semantic_scores, semantic_labels, weight,
                                                   self.ignore_label))
        # offset loss over the points of instances (reference softgroup.py:163-169: boolean indexing,
        # a host read of pos.sum() to branch on "no instance point").  Here without the read-back and
        # without the compaction: masked sum / count, 0 * sum when the count is 0 -- the same value
        # (summation order aside) and the same gradient.
        pos = (instance_labels != self.ignore_label)
        n_pos = pos.sum(dtype=torch.int32)
        diff = (pt_offsets - pt_offset_labels).abs()
        diff = torch.where(pos.unsqueeze(1), diff, torch.zeros((), dtype=diff.dtype, device=diff.device))
        losses['offset_loss'] = diff.sum() / n_pos.clamp(min=1)
        return losses

    @force_fp32(apply_to=('cls_scores', 'mask_scores', 'iou_scores'))
    def instance_loss(self, cls_scores, mask_scores, iou_scores, proposals_idx, proposals_offset,
                      instance_labels, instance_pointnum, instance_cls, instance_batch_idxs):
        if proposals_idx.size(0) == 0 or (instance_cls != self.ignore_label).sum() == 0:
            zero = mask_scores.sum() * 0
            return dict(cls_loss=cls_scores.sum() * 0, mask_loss=zero,
                        iou_score_loss=iou_scores.sum() * 0, num_pos=zero, num_neg=zero)
        dev = cls_scores.device
        tc = self.train_cfg
        pidx = proposals_idx[:, 1].int().to(dev).contiguous()
        poff = proposals_offset.to(dev).int().contiguous()
        instance_pointnum = instance_pointnum.int().contiguous()
        ious_on_cluster = ops.get_mask_iou_on_cluster(pidx, poff, instance_labels, instance_pointnum)

        fg = instance_cls != self.ignore_label
        labels = _assign_proposals(ious_on_cluster, instance_cls, fg, _cfg(tc, 'pos_iou_thr'),
                                   _cfg(tc, 'match_low_quality', False), _cfg(tc, 'min_pos_thr', 0),
                                   self.instance_classes)
        losses = dict(cls_loss=F.cross_entropy(cls_scores, labels))

        # mask loss on the score slice of the assigned class
        per_point_cls = labels[instance_batch_idxs.long()]
        rows = torch.arange(per_point_cls.size(0), device=dev)
        mask_sig = mask_scores.sigmoid()[rows, per_point_cls]
        mask_label = ops.get_mask_label(pidx, poff, instance_labels, instance_cls, instance_pointnum,
                                        ious_on_cluster, _cfg(tc, 'pos_iou_thr'))
        weight = (mask_label != -1).float()
        mask_label = torch.where(mask_label == -1., mask_label.new_full((), 0.5), mask_label)   # ignored points
        mask_loss = F.binary_cross_entropy(mask_sig, mask_label, weight=weight, reduction='sum')
        losses['mask_loss'] = mask_loss / (weight.sum() + 1)

        # IoU-score regression against the IoU of the predicted mask
        ious = ops.get_mask_iou_on_pred(pidx, poff, instance_labels, instance_pointnum,
                                        mask_sig.detach().contiguous())
        gt_ious, _ = torch.where(fg.unsqueeze(0), ious, ious.new_full((), -1.0)).max(1)
        rows = torch.arange(labels.size(0), device=dev)
        w = (labels < self.instance_classes).float()
        iou_loss = F.mse_loss(iou_scores[rows, labels], gt_ious, reduction='none')
        losses['iou_score_loss'] = (iou_loss * w).sum() / (w.sum() + 1)
        losses['num_pos'] = (labels < self.instance_classes).sum().float()
        losses['num_neg'] = (labels >= self.instance_classes).sum().float()
        return losses

    def parse_losses(self, losses):
        """-> (total loss, dict of python floats averaged over ranks).  The reference issues one
        all-reduce per scalar plus one for the key count (softgroup.py:281-295); here the key count
        and all scalars travel in ONE packed all-reduce over RCCL."""
        log_vars = OrderedDict()
        for name, value in losses.items():
            if isinstance(value, torch.Tensor):
                log_vars[name] = value.mean()
            elif isinstance(value, list):
                log_vars[name] = sum(v.mean() for v in value)
            else:
                raise TypeError(f'{name} is not a tensor or list of tensors')
        loss = sum(v for k, v in log_vars.items() if 'loss' in k)
        log_vars['loss'] = loss
        packed = torch.stack([v.detach().float() for v in log_vars.values()] +
                             [loss.new_tensor(float(len(log_vars)))])
        if dist.is_available() and dist.is_initialized():
            world = dist.get_world_size()
            dist.all_reduce(packed)
            assert int(packed[-1].item()) == len(log_vars) * world, \
                (f'loss log variables are different across GPUs!\nrank {dist.get_rank()} '
                 f'len(log_vars): {len(log_vars)} keys: ' + ','.join(log_vars.keys()))
            packed = packed / world
        vals = packed[:-1].tolist()
        return loss, OrderedDict((k, vals[i]) for i, k in enumerate(log_vars.keys()))


# ------------------------------------------------------------------------------------------------
def _assign_proposals(ious_on_cluster, instance_cls, fg, pos_iou_thr, match_low_quality, min_pos_thr,
                      background_label):
    """Proposal -> class label through the ground-truth assignment of the reference
    (softgroup.py:196-222): a proposal is positive for the GT of its largest IoU if that IoU reaches
    `pos_iou_thr`; with `match_low_quality` every GT whose best IoU reaches `min_pos_thr` also claims
    its best proposal (GTs in order, a later one overwrites an earlier one).  The reference drops the
    background GT columns by boolean indexing and writes the matches through boolean masks -- five
    host read-backs of a count.  Here the background columns stay and are masked to IoU -1 (they can
    never win a maximum; column order, hence every argmax tie-break, is unchanged) and the masked
    writes are torch.where: same labels, no read-back."""
    n_prop, n_gt = ious_on_cluster.shape
    dev = ious_on_cluster.device
    fg_ious = torch.where(fg.unsqueeze(0), ious_on_cluster, ious_on_cluster.new_full((), -1.0))
    max_iou, argmax_iou = fg_ious.max(1)
    assigned = torch.where(max_iou >= pos_iou_thr, argmax_iou, argmax_iou.new_full((), -1))
    if match_low_quality:
        gt_max, gt_arg = fg_ious.max(0)
        cand = torch.where(gt_max >= min_pos_thr, torch.arange(n_gt, device=dev), gt_arg.new_full((), -1))
        lowq = cand.new_full((n_prop, ), -1).scatter_reduce(0, gt_arg, cand, 'amax', include_self=True)
        assigned = torch.where(lowq >= 0, lowq, assigned)
    # classification: 0..K-1 foreground, K background
    return torch.where(assigned >= 0, instance_cls[assigned.clamp(min=0)],
                       instance_cls.new_full((), background_label))


def _cross_entropy(scores, labels, weight, ignore_index):
    """F.cross_entropy(scores, labels, weight=weight, ignore_index=ignore_index) (reference
    softgroup.py:159-160) as log_softmax + gather + masked mean: torch's nll_loss reduces [N] with a
    single workgroup (285 us for the 600 k points of a config-3 step), this form is three
    vectorised kernels.  Same value up to the order of the fp32 additions; same gradient."""
    logp = F.log_softmax(scores.float(), dim=1)
    valid = labels != ignore_index
    tgt = labels.clamp(min=0).unsqueeze(1)
    nll = -logp.gather(1, tgt).squeeze(1)
    w = valid.float() if weight is None else weight[tgt.squeeze(1)] * valid
    nll = torch.where(valid, nll, torch.zeros((), dtype=nll.dtype, device=nll.device))
    return (nll * w).sum() / w.sum()


def _take_rows(feats, index):
    """feats[index] through the HIP row-gather (devoxelize, softgroup.py:374,677-678); keeps
    autograd by falling back to torch indexing only when a gradient is required."""
    if feats.requires_grad and torch.is_grad_enabled():
        return feats[index.long()]
    from .. import _lib as L
    feats = feats.contiguous()
    if feats.dtype != torch.float32 or not feats.is_cuda:
        return feats[index.long()]
    index = index.contiguous()
    out = torch.empty((index.numel(), feats.shape[1]), dtype=torch.float32, device=feats.device)
    fn = L.lib().sg_gather_rows_i64idx_f32 if index.dtype == torch.int64 else L.lib().sg_gather_rows_f32
    if index.dtype not in (torch.int64, torch.int32):
        index = index.int()
    L.check(fn(L.ptr(feats), L.ptr(index), index.numel(), feats.shape[1], L.ptr(out), L.stream()),
            'sg_gather_rows')
    return out


def _runs_of_pairs(group, point, n_groups):
    """(group, point) pairs -> runs of consecutive points per group.
    Returns numpy (starts, lengths, bounds) with runs of group g in [bounds[g], bounds[g+1])."""
    if group.numel() == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(n_groups + 1, np.int64)
    big = int(point.max().item()) + 2
    key, _ = torch.sort(group * big + point)
    g, p = key // big, key % big
    new_run = torch.ones_like(key, dtype=torch.bool)
    new_run[1:] = (key[1:] != key[:-1] + 1) | (g[1:] != g[:-1])
    first = new_run.nonzero().view(-1)
    starts = p[first]
    ends = torch.cat([first[1:], first.new_tensor([key.numel()])])
    lens = ends - first
    run_group = g[first]
    bounds = torch.searchsorted(run_group, torch.arange(n_groups + 1, device=group.device))
    return starts.cpu().numpy(), lens.cpu().numpy(), bounds.cpu().numpy()
