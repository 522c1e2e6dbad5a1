"""Argument-casting decorators with the semantics of the reference's ``cuda_cast``
(softgroup/util/utils.py:157-173) and ``force_fp32`` (softgroup/util/fp16.py:24-66).
``force_fp32`` additionally up-casts bfloat16 (BASELINE config 3 trains in bf16; the reference
only knows torch.half)."""
import functools
import inspect
import os
import threading
import weakref
from collections import abc

import numpy as np
import torch

from ..spconv import SparseConvTensor

_LOW = (torch.half, torch.bfloat16)


def cuda_cast(func):

    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        args = [a.cuda(non_blocking=True) if isinstance(a, torch.Tensor) else a for a in args]
        kwargs = {k: (v.cuda(non_blocking=True) if isinstance(v, torch.Tensor) else v)
                  for k, v in kwargs.items()}
        return func(*args, **kwargs)

    return wrapper


def _to_fp32(x):
    if isinstance(x, torch.Tensor):
        return x.float() if x.dtype in _LOW else x
    if isinstance(x, SparseConvTensor):
        return x.replace_feature(x.features.float()) if x.features.dtype in _LOW else x
    if isinstance(x, (str, bytes)):
        return x
    if isinstance(x, abc.Mapping):
        return type(x)({k: _to_fp32(v) for k, v in x.items()})
    if isinstance(x, (list, tuple)):
        return type(x)(_to_fp32(v) for v in x)
    return x


def force_fp32(apply_to=None):
    """Cast the named arguments (all when None) to fp32 and run the method with autocast off."""

    def deco(method):
        names = inspect.getfullargspec(method).args

        @functools.wraps(method)
        def wrapper(*args, **kwargs):
            if not isinstance(args[0], torch.nn.Module):
                raise TypeError('@force_fp32 can only be used to decorate the method of nn.Module')
            chosen = names if apply_to is None else apply_to
            args = [_to_fp32(a) if n in chosen else a for n, a in zip(names, args)] + \
                list(args[len(names):])
            kwargs = {k: (_to_fp32(v) if k in chosen else v) for k, v in kwargs.items()}
            with torch.autocast('cuda', enabled=False):
                return method(*args, **kwargs)

        return wrapper

    return deco


_PAD = {}
_NP_DTYPE = {torch.float32: np.float32, torch.float64: np.float64, torch.float16: np.float16,
             torch.int64: np.int64, torch.int32: np.int32, torch.int16: np.int16, torch.int8: np.int8,
             torch.uint8: np.uint8, torch.bool: np.bool_}


def _pad_block(dev):
    t = _PAD.get(dev)
    if t is None:
        t = _PAD[dev] = torch.zeros(256, dtype=torch.uint8, device=dev)
    return t


def to_host_begin(tensors, stream=None):
    """First half of ``to_host``: pack the CUDA tensors with one cat kernel and start their ONE
    device-to-host copy -- on `stream` if given (it first waits for the current stream: a side stream
    lets the copy overlap whatever the caller enqueues next), else on the current stream.  Returns a
    handle for ``to_host_end``; the source tensors must stay referenced until then."""
    out, plan, total = {}, [], 0
    for k, t in tensors.items():
        if not t.is_cuda:
            out[k] = t.numpy()
            continue
        t = t.contiguous()
        nb = t.numel() * t.element_size()
        plan.append((k, t, total, nb))
        total += (nb + 255) // 256 * 256
    stage = done = None
    if plan:
        dev = plan[0][1].device
        cur = torch.cuda.current_stream(dev)
        run = cur
        if stream is not None and stream != cur:
            ready = torch.cuda.Event()
            ready.record(cur)
            stream.wait_event(ready)
            run = stream
        with torch.cuda.stream(run):
            pad = _pad_block(dev)
            pieces, at = [], 0
            for k, t, off, nb in plan:
                if off > at:
                    pieces.append(pad[:off - at])
                pieces.append(t.reshape(-1).view(torch.uint8))
                at = off + nb
            packed = torch.cat(pieces) if len(pieces) > 1 else pieces[0]
            stage = torch.empty(max(at, 1), dtype=torch.uint8, pin_memory=True)
            stage.copy_(packed, non_blocking=True)
            done = torch.cuda.Event()
            done.record(run)
    return list(tensors), out, plan, stage, done, (packed if plan else None)


# Page-locked staging blocks still referenced by result arrays.  A consumer that releases each scan's
# results before long (a pipelined evaluation, bench.py) gets zero-copy views of the pinned block; a
# loop that ACCUMULATES results (the reference's tools/test.py keeps every scan's arrays until the
# evaluation) would otherwise hold one ~12 MB block -- 16 MB after the host allocator's power-of-two
# rounding -- page-locked per scan: past `_PINNED_CAP` bytes of live blocks a scan's arrays are copied
# out to ordinary pageable memory (one memcpy) and its block goes back to the allocator at once.
_PINNED_CAP = int(os.environ.get('SG_PINNED_RESULTS_MB', '256')) << 20
_pinned_lock = threading.Lock()
_pinned_live = [0]


def pinned_result_bytes():
    """bytes of pinned staging blocks currently kept alive by result arrays (tests, diagnostics)"""
    with _pinned_lock:
        return _pinned_live[0]


def _unpin(nbytes):
    with _pinned_lock:
        _pinned_live[0] -= nbytes


def to_host_end(handle):
    keys, out, plan, stage, done, _packed = handle
    if plan:
        done.synchronize()
        nbytes = stage.numel()
        with _pinned_lock:
            keep_pinned = _pinned_live[0] + nbytes <= _PINNED_CAP
            if keep_pinned:
                _pinned_live[0] += nbytes
        block = stage.numpy()           # every result array is a view of `block` (numpy keeps it as their base)
        if keep_pinned:
            weakref.finalize(block, _unpin, nbytes)
        else:
            block = block.copy()        # pageable; the pinned block is released when `stage` goes
        for k, t, off, nb in plan:
            out[k] = block[off:off + nb].view(_NP_DTYPE[t.dtype]).reshape(tuple(t.shape))
    return {k: out[k] for k in keys}


def adopt_pinned_block(stage):
    """numpy view of a pinned uint8 tensor that a C call has filled (sg_scan_forward's dense results), under
    the same accounting as to_host_end: a view of the pinned block while fewer than SG_PINNED_RESULTS_MB of
    such blocks are alive, a pageable copy beyond"""
    nbytes = stage.numel()
    with _pinned_lock:
        keep_pinned = _pinned_live[0] + nbytes <= _PINNED_CAP
        if keep_pinned:
            _pinned_live[0] += nbytes
    block = stage.numpy()
    if keep_pinned:
        weakref.finalize(block, _unpin, nbytes)
        return block
    return block.copy()


def to_host(tensors):
    """dict of tensors -> dict of numpy arrays.  All CUDA tensors are packed on the device by ONE cat
    kernel and travel in ONE device-to-host copy into a pinned staging block (torch's caching host
    allocator recycles it; every copy node costs ~10 us of blit set-up whatever its size, and a scan
    returns eight dense arrays), with a single wait -- instead of one pageable, synchronous ``.cpu()``
    per tensor (the reference's result dicts are built that way, softgroup.py:323-360; round 3
    counted 31 copy nodes per scan).  The arrays are views of the staging block while less than
    SG_PINNED_RESULTS_MB (default 256) of such blocks are alive, pageable copies beyond that (see
    `_PINNED_CAP`): accumulating the results of a whole dataset does not accumulate page-locked memory."""
    return to_host_end(to_host_begin(tensors))
