"""Host side of the native U-Net executor (csrc/unet_exec.hip, include/softgroup_hip.h
``sg_unet_forward``): turns an ``input_conv`` / ``UBlock`` / ``output_layer`` module triple
(reference softgroup/model/softgroup.py:60-65,93-95; blocks.py:44-143) into the C descriptor --
packed weights, eval-mode BatchNorm as scale/shift -- and runs it with one call per forward.

Used for inference only (no autograd through it); the module path launches the same kernels and
remains the training path."""
import ctypes as C
import os
import operator
import threading

import torch
from torch import nn

from .. import _lib as L
from . import core


class _Block(C.Structure):
    _fields_ = [('cin', C.c_int), ('cout', C.c_int),
                ('bn1_scale', C.c_void_p), ('bn1_shift', C.c_void_p), ('w1', C.c_void_p),
                ('bn2_scale', C.c_void_p), ('bn2_shift', C.c_void_p), ('w2', C.c_void_p),
                ('w_i', C.c_void_p)]


class _Level(C.Structure):
    _fields_ = [('planes', C.c_int), ('n_blocks', C.c_int),
                ('blocks', C.POINTER(_Block)), ('tail', C.POINTER(_Block)),
                ('down_bn_scale', C.c_void_p), ('down_bn_shift', C.c_void_p), ('down_w', C.c_void_p),
                ('up_bn_scale', C.c_void_p), ('up_bn_shift', C.c_void_p), ('up_w', C.c_void_p)]


class _Desc(C.Structure):
    _fields_ = [('n_levels', C.c_int), ('levels', C.POINTER(_Level)),
                ('input_cin', C.c_int), ('input_w', C.c_void_p),
                ('out_bn_scale', C.c_void_p), ('out_bn_shift', C.c_void_p),
                ('input_cin_packed', C.c_int), ('arithmetic', C.c_int)]


_VERSION = operator.attrgetter('_version')
_DATA_PTR = torch.Tensor.data_ptr
_desc_lock = threading.Lock()
_ERR_WORKSPACE = -2     # SG_ERR_WORKSPACE (include/softgroup_hip.h)
# bf16 rows between the layers under bf16 autocast (arithmetic 3).  Off by default: measured at the SAME speed as bf16
# operands over fp32 rows (profiles/r06_rows16.txt: 64->64 x 77 k rows 35.3 against 34.8 us, backbone forward 1.22 against
# 1.20 ms of conv time) at slightly lower accuracy (4.4e-3 against 3.2e-3 of the fp32 output) -- with one MFMA per product
# the kernel waits neither for gather bytes nor for conversions.  Halves the activation memory of the forward.
ROWS16 = os.environ.get('SG_UNET_ROWS16', '0') != '0'
_arena = {}      # (device, stream) -> uint8 tensor, grow-only: concurrent scans on different
                 # streams never share an arena


def _get_arena(nbytes, device):
    key = (device, L.stream())
    t = _arena.get(key)
    if t is None or t.numel() < nbytes:
        # a quarter of headroom: scenes of a dataset differ by a few percent in voxel count, and an arena
        # that grows by exactly that much is reallocated (hundreds of MB through hipMalloc, milliseconds)
        # every time a slightly larger scene reaches this stream
        if t is not None:
            del _arena[key], t
        _arena[key] = t = torch.empty(int(nbytes) + int(nbytes) // 4, dtype=torch.uint8, device=device)
    return t


def release_stream(raw_stream):
    """drop the arena of a stream that is being retired (its forwards have finished)"""
    for key in [k for k in _arena if k[1] == raw_stream]:
        del _arena[key]


class UNetExecutor:
    """descriptor + cached device tensors for one (input_conv, unet, output_layer) triple"""

    def __init__(self, unet, input_conv=None, output_layer=None):
        self.unet, self.input_conv, self.output_layer = unet, input_conv, output_layer
        self._key = None
        self._keep = []
        self._desc = None

    # ---- eligibility: plain inference over the structure this executor understands
    def _tensors(self):
        """parameters and buffers of the three modules (the module tree is walked once; tensors
        replaced by .to()/.cuda() or updated in place show up through data_ptr / _version)"""
        ts = self.__dict__.get('_tensor_list')
        if ts is None:
            mods = [self.unet] + [m for m in (self.input_conv, self.output_layer) if m is not None]
            self._bns = [sub for m in mods for sub in m.modules() if isinstance(sub, nn.BatchNorm1d)]
            self._owners = [(dct, name) for m in mods for sub in m.modules()
                            for dct in (sub._parameters, sub._buffers) for name in dct
                            if dct[name] is not None]
            ts = self._tensor_list = True
        return [dct[name] for dct, name in self._owners]

    def usable(self, feats):
        if not (feats.is_cuda and feats.dtype == torch.float32):
            return False
        if torch.is_grad_enabled() and (feats.requires_grad or any(t.requires_grad for t in self._tensors())):
            return False
        if self.__dict__.get('_tensor_list') is None:
            self._tensors()
        # (through __dict__ / _buffers: nn.Module.__getattr__ costs 0.5 us per buffer read, 65 BatchNorms)
        for bn in self._bns:
            if bn.__dict__['training'] or bn._buffers['running_mean'] is None:
                return False
        return self._supported()

    def _supported(self):
        ok = getattr(self, '_ok', None)
        if ok is None:
            try:
                levels = self._walk(self.unet, dry=True)
                # sg_unet_forward moves rows as float4: every channel count a multiple of 4
                # (anything else takes the module path, whose scalar kernel covers it)
                ok = all(lv.planes % 4 == 0 for lv in levels)
            except (AssertionError, AttributeError, IndexError):
                ok = False
            self._ok = ok
        return ok

    # ---- descriptor
    def _bn(self, bn):
        s, b = core._bn_affine(bn)
        self._keep += [s, b]
        return s.data_ptr(), b.data_ptr()

    def _w(self, conv):
        w = conv.weight_packed()
        self._keep.append(w)
        return w.data_ptr()

    def _block(self, rb, dry):
        from ..model.blocks import ResidualBlock, Custom1x1Subm3d
        assert isinstance(rb, ResidualBlock)
        cb = list(rb.conv_branch._modules.values())
        assert len(cb) == 6 and isinstance(cb[0], nn.BatchNorm1d) and isinstance(cb[1], nn.ReLU) \
            and isinstance(cb[2], core.SubMConv3d) and isinstance(cb[3], nn.BatchNorm1d) \
            and isinstance(cb[4], nn.ReLU) and isinstance(cb[5], core.SubMConv3d)
        assert cb[2].bias is None and cb[5].bias is None
        ib = list(rb.i_branch._modules.values())
        assert len(ib) == 1
        one = ib[0] if isinstance(ib[0], Custom1x1Subm3d) else None
        assert one is not None or isinstance(ib[0], nn.Identity)
        assert one is None or one.bias is None
        b = _Block()
        b.cin, b.cout = cb[2].in_channels, cb[2].out_channels
        if dry:
            return b
        b.bn1_scale, b.bn1_shift = self._bn(cb[0])
        b.w1 = self._w(cb[2])
        b.bn2_scale, b.bn2_shift = self._bn(cb[3])
        b.w2 = self._w(cb[5])
        if one is not None:
            w = core.pack_weight(one.weight, one.out_channels, 1, one.in_channels, False)
            self._keep.append(w)
            b.w_i = w.data_ptr()
        return b

    def _walk(self, ub, dry=False):
        """-> list of _Level, outermost first"""
        from ..model.blocks import UBlock
        assert isinstance(ub, UBlock)
        lv = _Level()
        lv.planes = ub.nPlanes[0]
        blocks = list(ub.blocks._modules.values())
        lv.n_blocks = len(blocks)
        arr = (_Block * len(blocks))(*[self._block(b, dry) for b in blocks])
        self._keep.append(arr)
        lv.blocks = arr
        out = [lv]
        if len(ub.nPlanes) > 1:
            cv = list(ub.conv._modules.values())
            dc = list(ub.deconv._modules.values())
            assert len(cv) == 3 and isinstance(cv[0], nn.BatchNorm1d) and isinstance(cv[2], core.SparseConv3d)
            assert len(dc) == 3 and isinstance(dc[0], nn.BatchNorm1d) and isinstance(dc[2], core.SparseInverseConv3d)
            assert cv[2].bias is None and dc[2].bias is None
            tail = list(ub.blocks_tail._modules.values())
            assert len(tail) == len(blocks)
            tarr = (_Block * len(tail))(*[self._block(b, dry) for b in tail])
            self._keep.append(tarr)
            lv.tail = tarr
            if not dry:
                lv.down_bn_scale, lv.down_bn_shift = self._bn(cv[0])
                lv.down_w = self._w(cv[2])
                lv.up_bn_scale, lv.up_bn_shift = self._bn(dc[0])
                lv.up_w = self._w(dc[2])
            out += self._walk(ub.u, dry)
        return out

    def _state_key(self):
        """(cache epoch, identity, version counter and storage address of every parameter / buffer).
        This runs once per forward on ~400 tensors, so each component is one C-level map.  The address
        is part of the key because `param.data = x` -- which is what nn.Module._apply does for
        `.to()` / `.cuda()` / `.float()` on ANY submodule -- keeps both the Parameter's id and its
        version counter: without it a moved storage went unnoticed and the cached
        descriptor kept raw pointers into the old one."""
        ts = self._tensors()
        return (core.cache_epoch(), tuple(map(id, ts)), tuple(map(_VERSION, ts)), tuple(map(_DATA_PTR, ts))), ts

    def _descriptor(self):
        with _desc_lock:          # concurrent scans share the executor: build the descriptor once
            return self._descriptor_locked()

    def _descriptor_locked(self):
        key, ts = self._state_key()
        if self._desc is None or key != self._key:
            self._key_tensors = ts
            self._keep = []
            levels = self._walk(self.unet)
            larr = (_Level * len(levels))(*levels)
            d = _Desc()
            d.n_levels, d.levels = len(levels), larr
            if self.input_conv is not None:
                ic = list(self.input_conv._modules.values())
                assert len(ic) == 1 and isinstance(ic[0], core.SubMConv3d) and ic[0].bias is None
                d.input_cin = cin = ic[0].in_channels
                if cin % 16 == 0:
                    d.input_w = self._w(ic[0])
                else:
                    # weights zero-padded along Cin to a multiple of 16: the executor convolves a
                    # zero-padded copy of the features on the persistent MFMA kernel (the general
                    # kernel's scalar gather took 54 us on the 6 -> 32 conv of the bench scene)
                    cpk = (cin + 15) // 16 * 16
                    cout, kvol = ic[0].out_channels, 27
                    wp = torch.zeros((cout, kvol, cpk), dtype=torch.float32, device=ic[0].weight.device)
                    wp[:, :, :cin] = ic[0].weight.detach().float().reshape(cout, kvol, cin)
                    w = core.pack_weight(wp, cout, kvol, cpk, False)
                    self._keep.append(w)
                    d.input_w = w.data_ptr()
                    d.input_cin_packed = cpk
            if self.output_layer is not None:
                ol = list(self.output_layer._modules.values())
                assert len(ol) == 2 and isinstance(ol[0], nn.BatchNorm1d) and isinstance(ol[1], nn.ReLU)
                d.out_bn_scale, d.out_bn_shift = self._bn(ol[0])
            self._keep.append(larr)
            self._desc, self._key = d, key
            # packed weights / affines were just written on THIS thread's stream; other streams
            # (concurrent scans) will read them without an event in between: finish them now (once)
            torch.cuda.current_stream().synchronize()
        return self._desc

    # ---- run
    def __call__(self, x):
        """x: SparseConvTensor -> features [M, planes[0]] of the U-Net output (after output_layer)"""
        lib = L.lib()
        d = self._descriptor()
        if torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16:
            # bf16 autocast (a frozen backbone inside a training step, BASELINE config 3): the
            # convolutions take bf16 operands like the module path's gather_conv_bf16 does -- one
            # MFMA per product instead of the six of the fp32-accurate split; sums and stored
            # activations stay fp32.  (a per-call copy: the cached descriptor is shared by threads)
            # Arithmetic 3 (SG_UNET_ROWS16=0: arithmetic 2): the activations BETWEEN the layers are bf16 rows as well
            # -- what spconv keeps under the reference's autocast (tools/train.py:47) and what this package's module
            # path does (gather_conv_bf16 stores bf16) -- half the gather bytes, no conversion in the matrix loop.
            d = _Desc.from_buffer_copy(d)
            d.arithmetic = 3 if ROWS16 else 2
        feats = x.features.contiguous()
        idx = x.indices.contiguous()
        M = feats.shape[0]
        out = torch.empty((M, d.levels[0].planes), dtype=torch.float32, device=feats.device)
        if M == 0:
            return out
        # sg_unet_arena_bytes prices every level with all M voxels (a bound that always suffices);
        # real scenes shrink ~2-4x per level, so a quarter of it is tried first and the bound only
        # if the executor reports the arena too small (the cached arena only ever grows)
        full = lib.sg_unet_arena_bytes(C.byref(d), M)
        shape = (C.c_int32 * 3)(*x.spatial_shape)
        for nb in (max(full // 4, min(full, 64 << 20)), full):
            arena = _get_arena(nb, feats.device)
            rc = lib.sg_unet_forward(C.byref(d), L.ptr(feats), L.ptr(idx), M, shape, L.ptr(out),
                                     L.ptr(arena), arena.numel(), L.stream())
            if rc != _ERR_WORKSPACE or nb == full:
                L.check(rc, 'sg_unet_forward')
                break
        return out
