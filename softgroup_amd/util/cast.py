"""Argument-casting decorators with the semantics of the reference's ``cuda_cast``
(softgroup/util/utils.py:157-173) and ``force_fp32`` (softgroup/util/fp16.py:24-66).
``force_fp32`` additionally up-casts bfloat16 (BASELINE config 3 trains in bf16; the reference
only knows torch.half)."""
import functools
import inspect
from collections import abc

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


def to_host_end(handle):
    keys, out, plan, stage, done, _packed = handle
    if plan:
        done.synchronize()
        for k, t, off, nb in plan:
            out[k] = stage[off:off + nb].view(t.dtype).view(t.shape).numpy()
    return {k: out[k] for k in keys}


def to_host(tensors):
    """dict of tensors -> dict of numpy arrays.  All CUDA tensors are packed on the device by ONE cat
    kernel and travel in ONE device-to-host copy into a pinned staging block (torch's caching host
    allocator recycles it; every copy node costs ~10 us of blit set-up whatever its size, and a scan
    returns eight dense arrays), with a single wait -- instead of one pageable, synchronous ``.cpu()``
    per tensor (the reference's result dicts are built that way, softgroup.py:323-360; round 3
    counted 31 copy nodes per scan).  The arrays are views of the staging block, which they keep
    alive: release results you no longer need, or every scan page-locks a fresh block."""
    return to_host_end(to_host_begin(tensors))
