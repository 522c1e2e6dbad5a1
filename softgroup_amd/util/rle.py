"""Run-length encoding of 1-D binary masks in the wire format of the reference
(softgroup/util/rle.py:5-41): ``{'length': N, 'counts': 'start len start len ...'}`` with
1-based starts."""
import numpy as np


def rle_encode(mask):
    mask = np.asarray(mask)
    padded = np.concatenate([[0], mask, [0]])
    edges = np.flatnonzero(padded[1:] != padded[:-1]) + 1
    edges[1::2] -= edges[::2]
    return dict(length=mask.shape[0], counts=' '.join(str(x) for x in edges))


def rle_encode_runs(length, starts, lengths):
    """Same dict from 0-based run starts and lengths (already sorted)."""
    flat = np.empty(2 * len(starts), dtype=np.int64)
    flat[0::2] = np.asarray(starts, dtype=np.int64) + 1
    flat[1::2] = lengths
    return dict(length=int(length), counts=' '.join(map(str, flat.tolist())))


def rle_decode(rle):
    tok = rle['counts'].split()
    starts = np.asarray(tok[0::2], dtype=np.int64) - 1
    lens = np.asarray(tok[1::2], dtype=np.int64)
    mask = np.zeros(rle['length'], dtype=np.uint8)
    for lo, n in zip(starts, lens):
        mask[lo:lo + n] = 1
    return mask


def rle_encode_many(length, starts, lens, bounds, ends32=None):
    """RLE dicts of many masks at once through the native formatter (sg_rle_format_host).
    starts/lens: int64 arrays of all runs, bounds[g]..bounds[g+1] = runs of mask g.
    With ``ends32``: starts / ends32 are the int32 arrays of sg_instance_runs (lens ignored)."""
    import ctypes as C

    from .. import _lib as L
    if ends32 is None:
        starts = np.ascontiguousarray(starts, dtype=np.int64)
        lens = np.ascontiguousarray(lens, dtype=np.int64)
    else:
        starts = np.ascontiguousarray(starts, dtype=np.int32)
        lens = np.ascontiguousarray(ends32, dtype=np.int32)
    bounds = np.ascontiguousarray(bounds, dtype=np.int64)
    n = len(bounds) - 1
    offs = np.zeros(n + 1, dtype=np.int64)
    lib = L.lib()
    cap = int(lib.sg_rle_format_bound(len(starts), len(str(int(length) + 1))))
    buf = np.empty(cap, dtype=np.uint8)
    vp = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
    fmt = lib.sg_rle_format_host if ends32 is None else lib.sg_rle_format_runs_host
    L.check(fmt(vp(starts), vp(lens), vp(bounds), n, vp(buf), cap, vp(offs)), 'sg_rle_format_host')
    o = offs.tolist()
    text = buf[:o[n]].tobytes().decode('ascii')        # one decode, then plain string slices
    length = int(length)
    return [dict(length=length, counts=text[o[g]:o[g + 1]]) for g in range(n)]


_TEXT_STAGE = {}


def rle_text_to_dicts(length, text, offsets):
    """RLE dicts from the device text of sg_rle_format_device: ``text`` is the uint8 CUDA tensor,
    ``offsets`` the n+1 host offsets; mask g is text[offsets[g] : offsets[g+1] - 1] (the space
    after its last run dropped).  One pinned copy of the used part of the text, one decode."""
    import threading

    import torch
    n = len(offsets) - 1
    total = int(offsets[n])
    length = int(length)
    if total == 0:
        return [dict(length=length, counts='') for _ in range(n)]
    key = (threading.get_ident(), text.device)
    stage = _TEXT_STAGE.get(key)
    if stage is None or stage.numel() < total:
        stage = torch.empty(max(total * 2, 1 << 22), dtype=torch.uint8, pin_memory=True)
        _TEXT_STAGE[key] = stage
    stage[:total].copy_(text[:total], non_blocking=True)
    torch.cuda.current_stream(text.device).synchronize()
    s = str(memoryview(stage.numpy())[:total], 'ascii')
    return [dict(length=length, counts=s[offsets[g]:max(offsets[g + 1] - 1, offsets[g])])
            for g in range(n)]
