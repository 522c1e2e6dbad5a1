"""Host side of the native TRAINING executor (csrc/unet_train.hip, include/softgroup_hip.h
``sg_unet_train_forward`` / ``sg_unet_train_backward``): a U-Net of model/blocks.py in train() mode
as ONE autograd node -- forward and backward are one C call each instead of ~400 module / autograd
round trips per step (the DDP training step, reference tools/train.py:44-62 ->
softgroup/model/softgroup.py:113-150).  This module only marshals: the descriptor over the modules'
own parameter tensors, one flat gradient buffer whose slices become the ``.grad`` contributions, the
arena that keeps the activations alive until the backward has run."""
import ctypes as C

import torch
from torch import nn

from .. import _lib as L
from . import core

_ERR_WORKSPACE = -2     # SG_ERR_WORKSPACE


class _TBn(C.Structure):
    _fields_ = [('weight', C.c_void_p), ('bias', C.c_void_p), ('running_mean', C.c_void_p),
                ('running_var', C.c_void_p), ('momentum', C.c_float), ('eps', C.c_float),
                ('g_weight', C.c_void_p), ('g_bias', C.c_void_p)]


class _TConv(C.Structure):
    _fields_ = [('w', C.c_void_p), ('g_w', C.c_void_p)]


class _TBlock(C.Structure):
    _fields_ = [('cin', C.c_int), ('cout', C.c_int), ('bn1', _TBn), ('bn2', _TBn),
                ('c1', _TConv), ('c2', _TConv), ('ci', _TConv)]


class _TLevel(C.Structure):
    _fields_ = [('planes', C.c_int), ('n_blocks', C.c_int), ('blocks', C.POINTER(_TBlock)),
                ('tail', C.POINTER(_TBlock)), ('down_bn', _TBn), ('up_bn', _TBn),
                ('down', _TConv), ('up', _TConv)]


class _TDesc(C.Structure):
    _fields_ = [('n_levels', C.c_int), ('levels', C.POINTER(_TLevel)), ('input_cin', C.c_int),
                ('input', _TConv), ('out_bn', _TBn), ('arithmetic', C.c_int)]


class _Tape:
    """the C tape of one forward + the arena it points into; released if the backward never runs"""

    def __init__(self, owner, handle, arena, keep):
        self.owner, self.handle, self.arena, self.keep = owner, handle, arena, keep

    def take(self):
        h, self.handle = self.handle, None
        return h

    def done(self):
        if self.arena is not None:
            self.owner._free_arenas.append(self.arena)
            self.arena = None
        self.keep = None

    def __del__(self):
        try:
            if self.handle is not None:
                L.lib().sg_unet_train_release(C.c_void_p(self.handle))
                self.handle = None
            if self.arena is not None and self.owner is not None:
                self.owner._free_arenas.append(self.arena)
                self.arena = None
        except Exception:       # interpreter shutdown: the library or the owner may be gone already
            pass


class UNetTrainExecutor:
    """(input_conv, unet, output_layer) in train() mode -> features, differentiable"""

    def __init__(self, unet, input_conv=None, output_layer=None):
        self.unet, self.input_conv, self.output_layer = unet, input_conv, output_layer
        self._free_arenas = []
        self._plan = None

    # ---- structure walk: the modules in descriptor order
    def _collect(self):
        if self._plan is not None:
            return self._plan
        from ..model.blocks import Custom1x1Subm3d, ResidualBlock, UBlock
        bns, convs = [], []

        def block(rb):
            assert isinstance(rb, ResidualBlock)
            cb = list(rb.conv_branch._modules.values())
            assert len(cb) == 6 and isinstance(cb[0], nn.BatchNorm1d) and isinstance(cb[1], nn.ReLU) \
                and isinstance(cb[2], core.SubMConv3d) and isinstance(cb[3], nn.BatchNorm1d) \
                and isinstance(cb[4], nn.ReLU) and isinstance(cb[5], core.SubMConv3d)
            ib = list(rb.i_branch._modules.values())
            assert len(ib) == 1
            one = ib[0] if isinstance(ib[0], Custom1x1Subm3d) else None
            assert one is not None or isinstance(ib[0], nn.Identity)
            bns.extend([cb[0], cb[3]])
            convs.extend([cb[2], cb[5]] + ([one] if one is not None else []))
            return dict(bn1=cb[0], bn2=cb[3], c1=cb[2], c2=cb[5], ci=one)

        def level(ub):
            assert isinstance(ub, UBlock)
            lv = dict(planes=ub.nPlanes[0], blocks=[block(b) for b in ub.blocks._modules.values()])
            out = [lv]
            if len(ub.nPlanes) > 1:
                cv = list(ub.conv._modules.values())
                dc = list(ub.deconv._modules.values())
                assert len(cv) == 3 and isinstance(cv[0], nn.BatchNorm1d) and isinstance(cv[1], nn.ReLU) \
                    and isinstance(cv[2], core.SparseConv3d)
                assert len(dc) == 3 and isinstance(dc[0], nn.BatchNorm1d) and isinstance(dc[1], nn.ReLU) \
                    and isinstance(dc[2], core.SparseInverseConv3d)
                bns.extend([cv[0], dc[0]])
                convs.extend([cv[2], dc[2]])
                lv.update(down_bn=cv[0], down=cv[2], up_bn=dc[0], up=dc[2])
                out += level(ub.u)
                lv['tail'] = [block(b) for b in ub.blocks_tail._modules.values()]
                assert len(lv['tail']) == len(lv['blocks'])
            return out

        plan = dict(levels=level(self.unet), input=None, out_bn=None)
        if self.input_conv is not None:
            ic = list(self.input_conv._modules.values())
            assert len(ic) == 1 and isinstance(ic[0], core.SubMConv3d)
            plan['input'] = ic[0]
            convs.append(ic[0])
        if self.output_layer is not None:
            ol = list(self.output_layer._modules.values())
            assert len(ol) == 2 and isinstance(ol[0], nn.BatchNorm1d) and isinstance(ol[1], nn.ReLU)
            plan['out_bn'] = ol[0]
            bns.append(ol[0])
        plan['bns'], plan['convs'] = bns, convs
        plan['params'] = [c.weight for c in convs] + [t for b in bns for t in (b.weight, b.bias)]
        self._plan = plan
        return plan

    def usable(self, feats):
        """training mode over the structure this executor understands; everything else (eval-mode
        BatchNorm, cumulative-average momentum, biased convs, odd channel counts, CPU) takes the
        module path"""
        if not (feats.is_cuda and feats.dtype == torch.float32 and torch.is_grad_enabled()):
            return False
        if feats.shape[0] < 2:
            # a single row: torch.nn.BatchNorm1d in train() mode raises ("Expected more than 1 value
            # per channel"); the module path does, so it is the path that must see this input
            return False
        ok = self.__dict__.get('_ok')
        if ok is None:
            try:
                p = self._collect()
                ok = all(lv['planes'] % 4 == 0 for lv in p['levels']) and all(c.bias is None for c in p['convs'])
            except (AssertionError, AttributeError, IndexError):
                ok = False
            self._ok = ok
        if not ok:
            return False
        p = self._plan
        if not all(b.training and b.affine and b.track_running_stats and b.momentum is not None
                   and b.running_mean is not None for b in p['bns']):
            return False
        if not all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in p['params']):
            return False
        return feats.requires_grad or any(t.requires_grad for t in p['params'])

    # ---- descriptor over the parameters and a flat gradient buffer
    def _descriptor(self, grads):
        """grads: dict id(param) -> gradient tensor (or absent: no gradient wanted)"""
        p = self._collect()
        keep = []

        def gp(t):
            g = grads.get(id(t))
            return g.data_ptr() if g is not None else None

        def bn(m):
            s = _TBn()
            s.weight, s.bias = m.weight.data_ptr(), m.bias.data_ptr()
            s.running_mean, s.running_var = m.running_mean.data_ptr(), m.running_var.data_ptr()
            s.momentum, s.eps = float(m.momentum), float(m.eps)
            s.g_weight, s.g_bias = gp(m.weight), gp(m.bias)
            return s

        def conv(m):
            s = _TConv()
            if m is not None:
                s.w, s.g_w = m.weight.data_ptr(), gp(m.weight)
            return s

        def block(b):
            s = _TBlock()
            s.cin, s.cout = b['c1'].in_channels, b['c1'].out_channels
            s.bn1, s.bn2 = bn(b['bn1']), bn(b['bn2'])
            s.c1, s.c2, s.ci = conv(b['c1']), conv(b['c2']), conv(b['ci'])
            return s

        levels = []
        for lv in p['levels']:
            s = _TLevel()
            s.planes, s.n_blocks = lv['planes'], len(lv['blocks'])
            arr = (_TBlock * len(lv['blocks']))(*[block(b) for b in lv['blocks']])
            keep.append(arr)
            s.blocks = arr
            if 'tail' in lv:
                tarr = (_TBlock * len(lv['tail']))(*[block(b) for b in lv['tail']])
                keep.append(tarr)
                s.tail = tarr
                s.down_bn, s.up_bn = bn(lv['down_bn']), bn(lv['up_bn'])
                s.down, s.up = conv(lv['down']), conv(lv['up'])
            levels.append(s)
        larr = (_TLevel * len(levels))(*levels)
        keep.append(larr)
        d = _TDesc()
        d.n_levels, d.levels = len(levels), larr
        if p['input'] is not None:
            d.input_cin = p['input'].in_channels
            d.input = conv(p['input'])
        if p['out_bn'] is not None:
            d.out_bn = bn(p['out_bn'])
        if torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16:
            d.arithmetic = 2          # bf16 operands in the convolutions and their input gradients
        return d, keep

    def _arena(self, nbytes, device):
        best = None
        for i, t in enumerate(self._free_arenas):
            if t.device == device and t.numel() >= nbytes and (best is None or t.numel() < self._free_arenas[best].numel()):
                best = i
        if best is not None:
            return self._free_arenas.pop(best)
        self._free_arenas = [t for t in self._free_arenas if t.device != device]      # outgrown
        return torch.empty(int(nbytes), dtype=torch.uint8, device=device)

    def __call__(self, x):
        """x: SparseConvTensor -> features [M, planes[0]] after output_layer (differentiable)"""
        p = self._collect()
        return _UNetTrainFn.apply(x.features, self, x.indices, tuple(x.spatial_shape), *p['params'])


class _UNetTrainFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, feats, ex, indices, spatial_shape, *params):
        lib = L.lib()
        p = ex._collect()
        feats = feats.contiguous()
        idx = indices.contiguous()
        M = feats.shape[0]
        dev = feats.device
        # one flat buffer for every wanted parameter gradient; its slices are what backward returns
        wanted = [t for t, need in zip(params, ctx.needs_input_grad[4:]) if need]
        flat = torch.empty(sum(t.numel() for t in wanted), dtype=torch.float32, device=dev)
        grads, at = {}, 0
        for t in wanted:
            grads[id(t)] = flat[at:at + t.numel()].view(t.shape)
            at += t.numel()
        d, keep = ex._descriptor(grads)
        out = torch.empty((M, d.levels[0].planes), dtype=torch.float32, device=dev)
        shape = (C.c_int32 * 3)(*spatial_shape)
        need = C.c_size_t(0)
        tape = C.c_void_p(None)
        nbytes = lib.sg_unet_train_arena_hint(C.byref(d), M)
        for _ in range(3):
            arena = ex._arena(nbytes, dev)
            rc = lib.sg_unet_train_forward(C.byref(d), L.ptr(feats), L.ptr(idx), M, shape, L.ptr(out), L.ptr(arena),
                                           arena.numel(), C.byref(need), C.byref(tape), L.stream())
            if rc != _ERR_WORKSPACE:
                break
            nbytes = max(int(need.value), arena.numel() + (arena.numel() >> 1))
            del arena
        L.check(rc, 'sg_unet_train_forward')
        torch._foreach_add_([b.num_batches_tracked for b in p['bns']], 1)
        ctx.tape = _Tape(ex, tape.value, arena, (keep, d, feats, idx, params))
        ctx.grads = [grads.get(id(t)) for t in params]
        ctx.save_for_backward(out)          # (an in-place write to the result before backward is an error)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = L.lib()
        out, = ctx.saved_tensors
        tape = ctx.tape
        handle = tape.take()
        if handle is None:
            raise RuntimeError('UNetTrainExecutor: backward through the same forward twice '
                               '(retain_graph is not supported by the native training executor)')
        feats = tape.keep[2]
        g = g.contiguous().float()
        g_feats = torch.empty_like(feats) if ctx.needs_input_grad[0] else None
        rc = lib.sg_unet_train_backward(C.c_void_p(handle), L.ptr(g), L.ptr(g_feats) if g_feats is not None else None,
                                        L.stream())
        tape.done()
        L.check(rc, 'sg_unet_train_backward')
        # hand the slices over without keeping a reference: AccumulateGrad then takes each as the parameter's
        # .grad as it is; with the node still holding them it cloned every one (246 copies per step)
        grads, ctx.grads = ctx.grads, None
        return (g_feats, None, None, None) + tuple(grads)
