"""Data side of the hot path: the step right BEFORE ``forward_test``.

The reference collates on the CPU -- ``collate_fn`` concatenates the samples, runs the
single-threaded CPU ``voxelization_idx`` (data/custom.py:196-256, the call at :239) -- and then
``@cuda_cast`` copies every tensor of the batch dict to the GPU with blocking, pageable
``.cuda()`` calls (util/utils.py:157-173).  ``collate_device`` takes the SAME list of items
(what ``CustomDataset.__getitem__`` returns, data/custom.py:170-194) and returns the SAME batch
dict (keys, dtypes, values), but

  * the per-field concatenation lands in pinned staging buffers (grow-only, reused) and goes to
    the device with asynchronous copies on the current stream;
  * the voxel index (``voxel_coords``, ``v2p_map``, ``p2v_map``) is built on the device by
    ``ops.voxelization_idx`` (HIP path, bit-identical to the CPU op);
  * ``spatial_shape`` comes from the host copy of the coordinates, so nothing is read back.

Every tensor of the result is already resident: ``forward_test``'s ``cuda_cast`` is a no-op.
"""
import math
import os
import threading
import time

import numpy as np
import torch

from .. import _lib as L
from .. import ops

_STAGING_POOL, _STAGING_POOL_LOCK = [], threading.Lock()    # staging sets of finished prefetch_device loaders
_local = threading.local()   # per thread: {(key, dtype) -> [pinned tensor (grow-only), event of the
#                              last copy out of it]} -- loader threads never share a staging buffer


def _pinned(key, shape, dtype):
    n = int(np.prod(shape)) if len(shape) else 1
    staging = _local.__dict__.setdefault('staging', {})
    slot = staging.get((key, dtype))
    if slot is None or slot[0].numel() < n:
        slot = staging[(key, dtype)] = [torch.empty(max(n, 1), dtype=dtype).pin_memory(), None]
    if slot[1] is not None:
        slot[1].synchronize()        # the previous batch's async copy has left the buffer
    return slot, slot[0][:n].view(*shape)


def _to_device(key, parts, dtype, device):
    """torch.cat(parts) -> pinned staging -> async copy; returns the device tensor"""
    shape = (sum(p.shape[0] for p in parts), ) + tuple(parts[0].shape[1:])
    slot, host = _pinned(key, shape, dtype)
    # plain single-threaded numpy copies: a torch CPU op would wake the whole intra-op thread pool
    # (one thread per core) for a few MB and leave it spinning next to the launch thread
    # the copies run in C without the interpreter lock where they are a plain move or the float64 -> float32 cast
    # (libsoftgroup_hip.so: sg_host_copy_2d / sg_host_cast_f64_f32); anything else through numpy
    lib = L.lib()
    dst, row = host.numpy(), 0
    row_bytes = host[0:1].numel() * host.element_size() if shape[0] else 0
    for p in parts:
        n = p.shape[0]
        if n and p.dtype == dtype and p.is_contiguous():
            L.check(lib.sg_host_copy_2d(host.data_ptr() + row * row_bytes, row_bytes, p.data_ptr(), row_bytes, n, row_bytes),
                    'sg_host_copy_2d')
        elif n and p.dtype == torch.float64 and dtype == torch.float32 and p.is_contiguous():
            L.check(lib.sg_host_cast_f64_f32(host.data_ptr() + row * row_bytes, p.data_ptr(), p.numel()), 'sg_host_cast_f64_f32')
        elif n:
            dst[row:row + n] = p.numpy()          # casts when the dtypes differ
        row += n
    dev = torch.empty(shape, dtype=dtype, device=device)
    dev.copy_(host, non_blocking=True)
    if slot[1] is None:
        slot[1] = torch.cuda.Event()
    slot[1].record()
    return dev


def make_item(xyz, rgb, scale=50, semantic_label=None, instance_label=None, scan_id='scan'):
    """The item tuple of ``CustomDataset.__getitem__`` for test-time data (transform_test without
    the empirical rotation, data/custom.py:162-194): coordinates scaled and shifted to >= 0,
    instance statistics (getInstanceInfo, :115-135)."""
    n = xyz.shape[0]
    xyz_middle = np.ascontiguousarray(xyz, dtype=np.float32)
    v = xyz_middle.astype(np.float64) * scale
    v -= v.min(0)
    if instance_label is None:
        instance_label = np.full(n, -100, np.int64)
    if semantic_label is None:
        semantic_label = np.where(instance_label >= 0, 2 + instance_label % 18, 0).astype(np.int64)
    inst_ids = np.unique(instance_label[instance_label >= 0])
    pointnum = [int((instance_label == i).sum()) for i in inst_ids]
    inst_cls = [int(semantic_label[instance_label == i][0]) - 2 for i in inst_ids]
    center = np.zeros((n, 3), np.float32)
    for i in inst_ids:
        m = instance_label == i
        center[m] = xyz_middle[m].mean(0)
    pt_offset = np.where((instance_label >= 0)[:, None], center - xyz_middle, 0).astype(np.float32)
    return (scan_id, torch.from_numpy(v).long(), torch.from_numpy(xyz_middle),
            torch.from_numpy(np.ascontiguousarray(rgb)).float(), torch.from_numpy(semantic_label),
            torch.from_numpy(instance_label.astype(np.int64)), len(inst_ids), pointnum, inst_cls,
            torch.from_numpy(pt_offset))


def _fill_gaps(instance_label):
    """``CustomDataset.getCroppedInstLabel`` with every point kept (data/custom.py:136-143): while an
    id below the maximum is unused, the LAST id moves into it"""
    instance_label = np.array(instance_label, copy=True)
    j = 0
    while j < instance_label.max():
        if not (instance_label == j).any():
            instance_label[instance_label == instance_label.max()] = j
        j += 1
    return instance_label


def _rank_ids(instance_label):
    """``KITTIDataset.getCroppedInstLabel`` (data/kitti.py:76-88): ids replaced by their rank among
    the ids present, -100 kept (SemanticKITTI instance ids are 32-bit label words, not 0..n-1)"""
    ids = np.unique(instance_label)
    rank = np.cumsum(ids != -100) - 1
    out = np.where(ids == -100, -100, rank)[np.searchsorted(ids, instance_label)]
    # (dtype as np.vectorize infers it from the FIRST point: an unlabelled one maps to itself and
    # keeps the input's dtype, a labelled one to a Python int)
    keep = instance_label.size and instance_label.flat[0] == -100
    return out.astype(instance_label.dtype if keep else np.int64)


def kitti_labels(label, learning_map):
    """The label decoding of ``KITTIDataset`` (data/kitti.py:30-46 and 60-70): ``label`` = the int32
    words of a ``.label`` file, ``learning_map`` = the table of semantic-kitti.yaml.  Classes are
    re-ordered stuff 0..10, thing 11..18, unlabelled -100; the instance label is the WHOLE word for
    thing points and -100 for stuff points.  -> (semantic_label int64, instance_label int32)"""
    remap = {}
    for k, v in learning_map.items():
        remap[k] = -100 if v == 0 else (v + 10 if v < 9 else v - 9)
    label = np.array(label, dtype=np.int32, copy=True)
    sem = np.vectorize(remap.__getitem__)(label & 0xFFFF)
    label[sem <= 10] = -100
    return sem, label


def scan_item(xyz, rgb, semantic_label, instance_label, scale=50, scan_id='scan', cls_shift=2,
              x4_split=False, relabel='fill_gaps'):
    """What the reference's datasets return from ``__getitem__`` for one scan at TEST time
    (data/custom.py:170-194 with ``transform_test`` :162-168), value for value and dtype for dtype:

      * ``dataAugment(xyz, False, False, False, False)`` (:91-113) is NOT the identity: without
        ``rot`` the scene is turned by the fixed 0.35 pi about z ("empirically ... match the results
        from checkpoint"), in float64 (float32 points times a float64 matrix);
      * voxel coordinates = trunc(xyz_middle * scale - min) as int64; ``coords_float`` stays
        float64 here (``collate_fn`` casts it to float32, :231);
      * instance ids are made dense -- ``relabel='fill_gaps'``: ``getCroppedInstLabel`` (:136-143),
        the last id moves into a missing one; ``'rank'``: the SemanticKITTI override
        (data/kitti.py:76-88);
      * ``getInstanceInfo`` (:75-89): per-instance mean as float32, offsets ``float32 mean - float64
        point``; ``instance_cls`` = semantic label of the instance's first point minus ``cls_shift``
        (-100 stays): ScanNet 2 (data/scannetv2.py:27-31), STPLS3D 1 (data/stpls3d.py:10-15),
        SemanticKITTI 11 (data/kitti.py:117-121), S3DIS 0;
      * ``x4_split`` (S3DIS, data/s3dis.py:46-78): the rotated scene is cut into the four interleaved
        sub-clouds i, i+4, ...; each is shifted to ITS OWN minimum; everything is stored piece after
        piece and the voxel coordinates get the piece number as column 0 ([N, 4]).
    ``semantic_label`` / ``instance_label`` as the dataset's ``load`` yields them (any real dtype;
    unlabelled = -100)."""
    theta = 0.35 * math.pi
    m = np.matmul(np.eye(3), [[math.cos(theta), math.sin(theta), 0], [-math.sin(theta), math.cos(theta), 0],
                              [0, 0, 1]])
    xyz_middle = np.matmul(np.asarray(xyz), m)                       # float64
    rgb, semantic_label, instance_label = np.asarray(rgb), np.asarray(semantic_label), np.asarray(instance_label)
    if x4_split:
        pieces = [np.arange(i, xyz_middle.shape[0], 4) for i in range(4)]
        vs = []
        for b, piece in enumerate(pieces):
            vp = xyz_middle[piece] * scale
            vp -= vp.min(0)
            vs.append(np.concatenate([np.full((vp.shape[0], 1), b), vp], 1))
        v = np.concatenate(vs, 0)
        order = np.concatenate(pieces)
        xyz_middle, rgb = xyz_middle[order], rgb[order]
        semantic_label, instance_label = semantic_label[order], instance_label[order]
    else:
        v = xyz_middle * scale
        v -= v.min(0)
    assert relabel in ('fill_gaps', 'rank')
    instance_label = _fill_gaps(instance_label) if relabel == 'fill_gaps' else _rank_ids(instance_label)
    lab32 = instance_label.astype(np.int32)
    n_inst = max(int(lab32.max()) + 1, 0) if lab32.size else 0
    pt_mean = np.full((xyz_middle.shape[0], 3), -100.0, np.float32)
    pointnum, inst_cls = [], []
    for i in range(n_inst):
        sel = np.where(lab32 == i)
        pt_mean[sel] = xyz_middle[sel].mean(0)
        pointnum.append(sel[0].size)
        c = semantic_label[sel[0][0]]
        inst_cls.append(c - cls_shift if (cls_shift and c != -100) else c)
    pt_offset = pt_mean - xyz_middle                                 # float64
    return (scan_id, torch.from_numpy(v).long(), torch.from_numpy(xyz_middle),
            torch.from_numpy(rgb).float(), torch.from_numpy(semantic_label),
            torch.from_numpy(instance_label), n_inst, pointnum, inst_cls, torch.from_numpy(pt_offset))


def collate_x4_device(batch, min_spatial=128, device='cuda'):
    """Device-side version of the S3DIS test-time ``collate_fn`` (data/s3dis.py:80-115): ONE scan
    whose item already carries the four sub-clouds (``scan_item(..., x4_split=True)``).  Same dict:
    no ``coords`` key, ``batch_idxs`` all zero, ``batch_size`` 4, the instance lists wrapped in one
    more dimension -- quirks included, ``forward_test`` (x4_split) expects exactly this."""
    (scan_id, coord, coord_float, feat, semantic_label, instance_label, inst_num, inst_pointnum,
     inst_cls, pt_offset_label) = batch[0]
    dev = torch.device(device)
    d_coords = _to_device('coords', [coord], torch.int64, dev)
    out = {
        'scan_ids': [scan_id],
        'batch_idxs': torch.zeros(coord.shape[0], dtype=torch.int32, device=dev),
        'coords_float': _to_device('coords_float', [coord_float], torch.float32, dev),
        'feats': _to_device('feats', [feat], torch.float32, dev),
        'semantic_labels': _to_device('semantic_labels', [semantic_label], torch.int64, dev),
        'instance_labels': _to_device('instance_labels', [instance_label], torch.int64, dev),
        'instance_pointnum': torch.tensor([inst_pointnum], dtype=torch.int).to(dev, non_blocking=True),
        'instance_cls': torch.tensor([inst_cls], dtype=torch.long).to(dev, non_blocking=True),
        'pt_offset_labels': _to_device('pt_offset_labels', [pt_offset_label], torch.float32, dev),
        'spatial_shape': np.clip(coord.numpy().max(0)[1:] + 1, min_spatial, None),
        'batch_size': 4,
    }
    voxel_coords, v2p_map, p2v_map = ops.voxelization_idx(d_coords, 4)
    out.update(voxel_coords=voxel_coords, v2p_map=v2p_map, p2v_map=p2v_map)
    return out


def collate_device(batch, min_spatial=128, device='cuda'):
    """Device-side ``collate_fn`` (data/custom.py:196-256): same input items, same batch dict."""
    scan_ids, coords, coords_float, feats, sem, ins, pointnum, cls, offs = [], [], [], [], [], [], [], [], []
    total_inst, batch_id = 0, 0
    cmax = np.zeros(3, np.int64)
    for data in batch:
        if data is None:
            continue
        (scan_id, coord, coord_float, feat, semantic_label, instance_label, inst_num, inst_pointnum,
         inst_cls, pt_offset_label) = data
        if total_inst:
            lab = instance_label.numpy().copy()
            lab[lab != -100] += total_inst
            instance_label = torch.from_numpy(lab)
        total_inst += inst_num
        scan_ids.append(scan_id)
        coords.append(coord)
        if coord.numel():
            if coord.dtype == torch.int64 and coord.is_contiguous() and coord.dim() == 2 and coord.shape[1] <= 8:
                one = np.empty(coord.shape[1], np.int64)      # (C loop, no interpreter lock: numpy's axis-0 max of a
                L.check(L.lib().sg_host_colmax_i64(coord.data_ptr(), coord.shape[0], coord.shape[1],      # [150 000, 3]
                                                   one.ctypes.data), 'sg_host_colmax_i64')          # array is 1.3 ms)
                cmax = np.maximum(cmax, one)
            else:
                cmax = np.maximum(cmax, coord.numpy().max(0))
        coords_float.append(coord_float)
        feats.append(feat)
        sem.append(semantic_label)
        ins.append(instance_label)
        pointnum.extend(inst_pointnum)
        cls.extend(inst_cls)
        offs.append(pt_offset_label)
        batch_id += 1
    assert batch_id > 0, 'empty batch'
    dev = torch.device(device)
    # coords [N, 1+3] with the batch index in column 0, assembled directly in the staging buffer
    n_total = sum(c.shape[0] for c in coords)
    slot, host = _pinned('coords', (n_total, 4), torch.int64)
    lib = L.lib()
    dst, row = host.numpy(), 0
    for b, c in enumerate(coords):
        n = c.shape[0]
        if n and c.dtype == torch.int64 and c.is_contiguous() and c.shape[1] == 3:      # (no interpreter lock held)
            L.check(lib.sg_host_fill_i64_strided(host.data_ptr() + row * 32, 4, n, b), 'sg_host_fill_i64_strided')
            L.check(lib.sg_host_copy_2d(host.data_ptr() + row * 32 + 8, 32, c.data_ptr(), 24, n, 24), 'sg_host_copy_2d')
        elif n:
            dst[row:row + n, 0] = b
            dst[row:row + n, 1:] = c.numpy()
        row += n
    d_coords = torch.empty((n_total, 4), dtype=torch.int64, device=dev)
    d_coords.copy_(host, non_blocking=True)
    if slot[1] is None:
        slot[1] = torch.cuda.Event()
    slot[1].record()
    out = {
        'scan_ids': scan_ids,
        'coords': d_coords,
        'batch_idxs': d_coords[:, 0].int(),
        'coords_float': _to_device('coords_float', coords_float, torch.float32, dev),
        'feats': _to_device('feats', feats, torch.float32, dev),
        'semantic_labels': _to_device('semantic_labels', sem, torch.int64, dev),
        'instance_labels': _to_device('instance_labels', ins, torch.int64, dev),
        'instance_pointnum': torch.tensor(pointnum, dtype=torch.int).to(dev, non_blocking=True),
        'instance_cls': torch.tensor(cls, dtype=torch.long).to(dev, non_blocking=True),
        'pt_offset_labels': _to_device('pt_offset_labels', offs, torch.float32, dev),
        'spatial_shape': np.clip(cmax + 1, min_spatial, None),
        'batch_size': batch_id,
    }
    voxel_coords, v2p_map, p2v_map = ops.voxelization_idx(d_coords, batch_id)   # device path
    out.update(voxel_coords=voxel_coords, v2p_map=v2p_map, p2v_map=p2v_map)
    return out


def prefetch_device(batches, collate=None, depth=1, device='cuda', workers=1):
    """Generator over device-resident batch dicts with the NEXT batches' collate -- the copies into pinned
    staging, the asynchronous H2D transfers and the device voxel index -- running on loader threads with
    their own streams while the consumer works on the current one: what the reference gets from DataLoader workers
    (data/__init__.py:28-47 building data/custom.py:196-256's ``collate_fn`` ahead of the test loop,
    tools/test.py:145).  ``batches`` yields lists of dataset items; ``collate`` defaults to collate_device.
    ``workers`` loader threads take the batches round-robin (batch i on thread i % workers, each up to ``depth``
    batches ahead) and the generator yields them IN ORDER: with scans in flight on a busy GPU one collate takes
    longer than on an idle one (its voxel-index kernels and their read-back queue behind the scans' kernels), two
    loaders hide that.  The consumer's current stream is made to wait for the loader stream's work (an event per
    batch); tensors are registered with the consumer's stream so the caching allocator does not hand them out early."""
    import queue
    collate = collate or collate_device
    dev = torch.device(device)
    workers = max(1, int(workers))
    stop = threading.Event()
    src_lock = threading.Lock()
    src = iter(batches)
    state = {'next': 0}
    qs = [queue.Queue(maxsize=max(1, int(depth))) for _ in range(workers)]
    # the loader threads' pinned staging buffers outlive them (page-locking 11 MB costs tens of ms: a new set per
    # generator made the first measurement of this path 32 ms per scan): taken from / returned to a module pool
    stagings = []
    with _STAGING_POOL_LOCK:
        for _ in range(workers):
            stagings.append(_STAGING_POOL.pop() if _STAGING_POOL else {})

    def loader(w):
        _local.staging = stagings[w]
        q = qs[w]
        try:
            with torch.cuda.device(dev):
                # (SG_LOADER_PRIORITY=1, developer knob: a high-priority stream for the collate's dozen small kernels,
                #  which queue behind the scans' persistent conv kernels on a busy GPU.  Measured worse: scans in
                #  flight fed from the host 4.6 against 3.85 ms/scan, one scan with the next prefetched 5.95 against 5.25)
                prio = -1 if os.environ.get('SG_LOADER_PRIORITY', '0') == '1' else 0
                side = torch.cuda.Stream(priority=prio)
                with torch.cuda.stream(side), torch.no_grad():
                    while not stop.is_set():
                        # batch i belongs to thread i % workers: take from the source in order, under the lock
                        with src_lock:
                            if state['next'] % workers != w:
                                mine = None
                            else:
                                try:
                                    mine = next(src)
                                except StopIteration:
                                    mine = StopIteration
                                state['next'] += 1
                        if mine is None:
                            time.sleep(0.0002)
                            continue
                        if mine is StopIteration:
                            break
                        out = collate(mine, device=device)
                        ev = torch.cuda.Event()
                        ev.record(side)
                        q.put((out, ev))
            q.put(None)
        except BaseException as e:      # noqa: BLE001 -- handed to the consumer
            q.put(e)

    threads = [threading.Thread(target=loader, args=(w, ), name=f'sg-prefetch-{w}', daemon=True) for w in range(workers)]
    for t in threads:
        t.start()
    try:
        k = 0
        while True:
            got = qs[k % workers].get()
            if got is None:
                return
            if isinstance(got, BaseException):
                raise got
            out, ev = got
            cur = torch.cuda.current_stream(dev)
            cur.wait_event(ev)
            for v in out.values():
                if isinstance(v, torch.Tensor) and v.is_cuda:
                    v.record_stream(cur)
            yield out
            k += 1
    finally:
        stop.set()
        for t in threads:
            while t.is_alive():
                for q in qs:
                    try:
                        q.get_nowait()
                    except queue.Empty:
                        pass
                t.join(0.01)
        for staging in stagings:      # (copies out of the buffers have finished before they are reused)
            for slot in staging.values():
                if slot[1] is not None:
                    slot[1].synchronize()
        with _STAGING_POOL_LOCK:
            _STAGING_POOL.extend(stagings)
