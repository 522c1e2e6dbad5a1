"""Host-side mirror of the ``spconv.pytorch`` subset the reference model uses.

Reference call sites (paths relative to the reference repository):
  SparseConvTensor      softgroup/model/softgroup.py:120,307,388,671,706; blocks.py:37,73,134
  SubMConv3d            softgroup/model/softgroup.py:61-62; blocks.py:57-70
  SparseConv3d          blocks.py:31 (base of Custom1x1Subm3d), 101-107 (k2, s2)
  SparseInverseConv3d   blocks.py:114-119
  SparseSequential      softgroup/model/softgroup.py:60,65,74; blocks.py:50-55,96-129
  modules.SparseModule  blocks.py:5,44

Parameters keep spconv 2.x naming and layout -- ``weight`` of shape [Cout, kD, kH, kW, Cin]
("OKKKI", tools/convert_checkpoint.py:17-19), optional ``bias`` -- so reference checkpoints load.
All arithmetic runs in libsoftgroup_hip.so (spconv_rulebook.hip, spconv_conv.hip).
"""
import ctypes as C
import math
import threading
from collections import OrderedDict

import torch
from torch import nn

from .. import _lib as L


# ------------------------------------------------------------------------------------------------
class SparseConvTensor:
    """features [M,C] float, indices int32 [M,4] = (batch, d0, d1, d2), spatial_shape, batch_size.
    ``indice_dict`` caches rulebooks by ``indice_key`` and is shared by tensors derived from it."""

    def __init__(self, features, indices, spatial_shape, batch_size, grid=None, voxel_num=None,
                 indice_dict=None, benchmark=False):
        assert features.dim() == 2 and indices.dim() == 2 and indices.shape[1] == 4
        assert indices.dtype == torch.int32, 'indices must be int32 (spconv contract)'
        self.features = features
        self.indices = indices
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = {} if indice_dict is None else indice_dict
        self.grid = grid
        self.voxel_num = voxel_num
        self.benchmark = benchmark

    def replace_feature(self, feature):
        new = SparseConvTensor(feature, self.indices, self.spatial_shape, self.batch_size, self.grid,
                               self.voxel_num, self.indice_dict, self.benchmark)
        return new

    @property
    def spatial_size(self):
        return int(torch.tensor(self.spatial_shape).prod())

    def find_indice_pair(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key)

    def dense(self, channels_first=True):
        shape = [self.batch_size] + self.spatial_shape + [self.features.shape[1]]
        out = self.features.new_zeros(shape)
        idx = self.indices.long()
        out[idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]] = self.features
        return out.permute(0, 4, 1, 2, 3).contiguous() if channels_first else out


# ------------------------------------------------------------------------------------------------
# rulebooks
# ------------------------------------------------------------------------------------------------
_PLAN_HIST_WORDS = 40     # SG_PLAN_HIST_WORDS (include/softgroup_hip.h)


class _Plan:
    """gather table + mask-sorted tile plan for the implicit-GEMM kernel"""
    __slots__ = ('nbr', 'order', 'tile_mask', 'nbr_tiles', 'num_out', 'kvol', '_pairs', '_nbr_t')

    def __init__(self, nbr, num_out, kvol):
        lib = L.lib()
        dev = nbr.device
        self.nbr, self.num_out, self.kvol = nbr, num_out, kvol
        self._pairs = None
        self._nbr_t = None
        nt = (num_out + 31) // 32
        self.order = torch.empty(nt * 32, dtype=torch.int32, device=dev)
        self.tile_mask = torch.empty(nt + _PLAN_HIST_WORDS, dtype=torch.int32, device=dev)   # masks + histogram
        self.nbr_tiles = torch.empty(nt * 32 * kvol, dtype=torch.int32, device=dev)
        if num_out:
            nb = lib.sg_spconv_plan_workspace_bytes(num_out)
            ws = L.workspace(nb, dev)
            L.check(lib.sg_spconv_plan(L.ptr(nbr), num_out, kvol, L.ptr(self.order),
                                       L.ptr(self.tile_mask), L.ptr(self.nbr_tiles), L.ptr(ws), nb,
                                       L.stream()),
                    'sg_spconv_plan')


    @property
    def nbr_t(self):
        """gather table transposed to [K, M] (weight gradient only; built on first use)"""
        if self._nbr_t is None:
            t = torch.empty((self.kvol, self.num_out), dtype=torch.int32, device=self.nbr.device)
            L.check(L.lib().sg_spconv_transpose_table(L.ptr(self.nbr), self.num_out, self.kvol,
                                                      L.ptr(t), L.stream()), 'sg_spconv_transpose_table')
            self._nbr_t = t
        return self._nbr_t


class SubMRule:
    def __init__(self, indices, spatial_shape):
        lib = L.lib()
        M = indices.shape[0]
        dev = indices.device
        nbr = torch.empty((M, 27), dtype=torch.int32, device=dev)
        if M:
            nb = lib.sg_spconv_hash_workspace_bytes(M)
            ws = L.workspace(nb, dev)
            shape = (C.c_int32 * 3)(*spatial_shape)
            L.check(lib.sg_spconv_subm_rulebook(L.ptr(indices), M, shape, L.ptr(nbr), L.ptr(ws), nb,
                                                L.stream()), 'sg_spconv_subm_rulebook')
        self.num_rows = M
        self.plan = _Plan(nbr, M, 27)


class DownRule:
    """SparseConv3d(k=2, s=2) pairs; also serves the paired SparseInverseConv3d."""

    def __init__(self, indices, spatial_shape, batch_size):
        lib = L.lib()
        M = indices.shape[0]
        dev = indices.device
        self.in_indices = indices
        self.in_spatial_shape = list(spatial_shape)
        self.out_spatial_shape = [s // 2 for s in spatial_shape]   # floor((D-2)/2)+1
        self.batch_size = batch_size
        self.in2out = torch.empty(M, dtype=torch.int32, device=dev)
        meta = torch.zeros(2, dtype=torch.int32, device=dev)
        nb = lib.sg_spconv_hash_workspace_bytes(M)
        ws = L.workspace(nb, dev)
        shape = (C.c_int32 * 3)(*spatial_shape)
        st = L.stream()
        L.check(lib.sg_spconv_down_build(L.ptr(indices), M, shape, L.ptr(self.in2out), L.ptr(meta),
                                         L.ptr(ws), nb, st), 'sg_spconv_down_build')
        m_out = int(meta[0].item())
        self.num_in, self.num_out = M, m_out
        self.out_indices = torch.empty((m_out, 4), dtype=torch.int32, device=dev)
        child = torch.empty((m_out, 8), dtype=torch.int32, device=dev)
        L.check(lib.sg_spconv_down_fill(L.ptr(indices), M, L.ptr(self.in2out), m_out,
                                        L.ptr(self.out_indices), L.ptr(child), L.ptr(ws), nb, st),
                'sg_spconv_down_fill')
        self.plan = _Plan(child, m_out, 8)
        self._inv_plan = None

    @property
    def inv_plan(self):
        if self._inv_plan is None:
            lib = L.lib()
            inv = torch.empty((self.num_in, 8), dtype=torch.int32, device=self.in2out.device)
            if self.num_in:
                L.check(lib.sg_spconv_inverse_rulebook(L.ptr(self.in_indices), L.ptr(self.in2out),
                                                       self.num_in, L.ptr(inv), L.stream()),
                        'sg_spconv_inverse_rulebook')
            self._inv_plan = _Plan(inv, self.num_in, 8)
        return self._inv_plan


# ------------------------------------------------------------------------------------------------
# the conv operator
# ------------------------------------------------------------------------------------------------
class ConvProfiler:
    """Optional per-launch timing of the conv kernel with HIP events on the launch stream, plus the
    algorithmic byte/flop counts of SURVEY 8(d):  B_gs = P*Cin*4 + M_out*Cout*4 + 8*P,
    flops = 2*P*Cin*Cout  (P = active (in,out) pairs of the layer).  Used by bench.py."""

    def __init__(self):
        self.records = []      # (start_event, end_event, bytes, flops, tag)

    def summary(self):
        torch.cuda.synchronize()
        ms = sum(s.elapsed_time(e) for s, e, _, _, _ in self.records)
        return dict(launches=len(self.records), ms=ms,
                    bytes=sum(r[2] for r in self.records), flops=sum(r[3] for r in self.records))


PROFILER = None      # set to a ConvProfiler to record


def _plan_pairs(plan):
    if getattr(plan, '_pairs', None) is None:
        plan._pairs = int((plan.nbr >= 0).sum().item())
    return plan._pairs


def pack_weight(w, cout, kvol, cin, src_is_kio):
    """[Cout,K,Cin] (src_is_kio=False) or [K,Cin,Cout] (True) -> the conv kernel's "k8" layout"""
    lib = L.lib()
    src = w.detach().float().contiguous()
    out = torch.empty(lib.sg_spconv_packed_weight_elems(kvol, cin, cout), dtype=torch.float32,
                      device=w.device)
    L.check(lib.sg_spconv_pack_weight(L.ptr(src), cout, kvol, cin, 1 if src_is_kio else 0,
                                      L.ptr(out), L.stream()), 'sg_spconv_pack_weight')
    return out


def bn_relu(feats, scale, shift):
    """relu(feats * scale + shift) over [M, C] rows, one elementwise HIP kernel"""
    feats = feats.contiguous()
    out = torch.empty_like(feats)
    L.check(L.lib().sg_bn_relu_f32(L.ptr(feats), L.ptr(scale), L.ptr(shift), feats.shape[0],
                                   feats.shape[1], 1, L.ptr(out), L.stream()), 'sg_bn_relu_f32')
    return out


def gather_conv(features, plan, w_k8, cout, post_scale=None, post_shift=None, residual=None):
    """out[j] = post(residual[j] + sum_k features[nbr[j,k]] @ W[k])   (fp32, HIP);
    post = relu(x*post_scale + post_shift) when given"""
    lib = L.lib()
    prof = PROFILER
    if prof is not None:
        P = _plan_pairs(plan)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    assert features.is_cuda and features.dtype == torch.float32 and features.is_contiguous()
    cin = features.shape[1]
    out = torch.empty((plan.num_out, cout), dtype=torch.float32, device=features.device)
    if residual is not None:
        assert residual.shape == out.shape and residual.is_contiguous()
    nb = lib.sg_spconv_conv_workspace_bytes(plan.num_out, cout)
    ws = L.workspace(nb, features.device) if nb > 256 else None
    if prof is not None:
        ev0.record()
    L.check(lib.sg_spconv_gather_conv_f32(
        L.ptr(features), features.shape[0], L.ptr(plan.nbr), plan.num_out, plan.kvol, cin, cout,
        L.ptr(w_k8), L.ptr(post_scale), L.ptr(post_shift), L.ptr(residual), None, None, None,
        L.ptr(plan.order),
        L.ptr(plan.tile_mask), L.ptr(plan.nbr_tiles), L.ptr(out), L.ptr(ws),
        nb if ws is not None else 0, L.stream()),
        'sg_spconv_gather_conv_f32')
    if prof is not None:
        ev1.record()
        prof.records.append((ev0, ev1, P * cin * 4 + plan.num_out * cout * 4 + 8 * P,
                             2 * P * cin * cout, (plan.kvol, cin, cout, plan.num_out)))
    return out


def pack_weight_bf16(w, cout, kvol, cin, src_is_kio):
    """fp32 master weight -> the bf16 kernel's packed layout (rounded to nearest even)"""
    lib = L.lib()
    src = w.detach().float().contiguous()
    out = torch.empty(lib.sg_spconv_packed_weight_elems_bf16(kvol, cin, cout), dtype=torch.bfloat16,
                      device=w.device)
    L.check(lib.sg_spconv_pack_weight_bf16(L.ptr(src), cout, kvol, cin, 1 if src_is_kio else 0,
                                           L.ptr(out), L.stream()), 'sg_spconv_pack_weight_bf16')
    return out


def gather_conv_bf16(features, plan, w_k8, cout):
    """out[j] = bf16(sum_k features[nbr[j,k]] @ W[k]): bf16 operands, fp32 accumulation (HIP, MFMA
    bf16); the autocast path of training (BASELINE config 3)."""
    lib = L.lib()
    assert features.is_cuda and features.dtype == torch.bfloat16 and features.is_contiguous()
    cin = features.shape[1]
    out = torch.empty((plan.num_out, cout), dtype=torch.bfloat16, device=features.device)
    nb = lib.sg_spconv_conv_bf16_workspace_bytes(plan.num_out, cout)
    ws = L.workspace(nb, features.device) if nb > 256 else None
    L.check(lib.sg_spconv_gather_conv_bf16(
        L.ptr(features), features.shape[0], L.ptr(plan.nbr), plan.num_out, plan.kvol, cin, cout,
        L.ptr(w_k8), L.ptr(plan.order), L.ptr(plan.tile_mask), L.ptr(plan.nbr_tiles), L.ptr(out),
        L.ptr(ws), nb if ws is not None else 0, L.stream()), 'sg_spconv_gather_conv_bf16')
    return out


def conv_wgrad(feats, g, plan, cin, cout):
    """dW [K, Cin, Cout] fp32 = sum_j feats[nbr[j,k]]^T g[j]; feats / g fp32 or bf16.
    Deterministic (fixed-order chunk sums)."""
    lib = L.lib()
    assert feats.is_contiguous() and g.is_contiguous()
    assert feats.dtype in (torch.float32, torch.bfloat16) and g.dtype in (torch.float32, torch.bfloat16)
    kvol = plan.kvol
    dw = torch.empty((kvol, cin, cout), dtype=torch.float32, device=g.device)
    nb = lib.sg_spconv_wgrad_workspace_bytes(plan.num_out, kvol, cin, cout)
    ws = L.workspace(nb, g.device)
    L.check(lib.sg_spconv_wgrad(L.ptr(feats), int(feats.dtype == torch.bfloat16), L.ptr(g),
                                int(g.dtype == torch.bfloat16), L.ptr(plan.nbr_t), plan.num_out, kvol,
                                cin, cout, L.ptr(dw), L.ptr(ws), nb, L.stream()), 'sg_spconv_wgrad')
    return dw


class _GatherConvFn(torch.autograd.Function):
    """Differentiable sparse conv.  forward = gather_conv (fp32) or gather_conv_bf16 (features in
    bf16: the autocast path); input gradient = the same kernel on the transposed gather table with
    the transposed weights (packed once per weight version, ``packed(transposed, bf16)``); weight
    gradient = sg_spconv_wgrad (fp32 sums, deterministic)."""

    @staticmethod
    def forward(ctx, feats, weight, packed, plan, bwd_plan):
        ctx.save_for_backward(feats, weight)
        ctx.plan, ctx.bwd_plan, ctx.packed = plan, bwd_plan, packed
        if feats.dtype == torch.bfloat16:
            return gather_conv_bf16(feats, plan, packed(False, True), weight.shape[0])
        return gather_conv(feats, plan, packed(False, False), weight.shape[0])

    @staticmethod
    def backward(ctx, g):
        feats, weight = ctx.saved_tensors
        plan, bwd_plan = ctx.plan, ctx.bwd_plan
        cout, cin = weight.shape[0], weight.shape[-1]
        low = feats.dtype == torch.bfloat16
        g = g.contiguous().to(feats.dtype)
        g_feats = g_weight = None
        if ctx.needs_input_grad[0]:
            if low:
                g_feats = gather_conv_bf16(g, bwd_plan, ctx.packed(True, True), cin)
            else:
                g_feats = gather_conv(g, bwd_plan, ctx.packed(True, False), cin)
        if ctx.needs_input_grad[1]:
            dw = conv_wgrad(feats, g, plan, cin, cout)
            g_weight = dw.permute(2, 0, 1).reshape(weight.shape).to(weight.dtype)
        return g_feats, g_weight, None, None, None


# ---- derived-tensor caches (packed conv weights, eval-mode BatchNorm affines, executor
# descriptors) are keyed on (tensor._version, data_ptr) of their sources PLUS this epoch.
# load_state_dict / optimizer steps / .to() change the key by themselves; writes through
# ``param.data`` (EMA copies, custom initialisers) do NOT bump ``_version`` -- after such writes
# call ``invalidate_caches()`` (SoftGroup.train(), ._apply() and .load_state_dict() do it too).
_CACHE_EPOCH = [0]


def invalidate_caches():
    """Drop every cached packed weight / BatchNorm affine / executor descriptor (they are rebuilt
    on the next forward).  Needed after in-place writes through ``tensor.data``."""
    _CACHE_EPOCH[0] += 1


def cache_epoch():
    return _CACHE_EPOCH[0]


class SparseModule(nn.Module):
    """marker base class: SparseSequential hands these the SparseConvTensor itself"""
    pass


def is_spconv_module(module):
    return isinstance(module, SparseModule)


def _triple(v):
    return tuple(v) if isinstance(v, (list, tuple)) else (v, v, v)


class SparseConvolution(SparseModule):

    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0,
                 dilation=1, groups=1, bias=True, subm=False, output_padding=0, transposed=False,
                 inverse=False, indice_key=None, algo=None, fp32_accum=None, name=None):
        super().__init__()
        assert ndim == 3 and groups == 1
        self.ndim = ndim
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _triple(kernel_size), _triple(stride)
        self.padding, self.dilation = _triple(padding), _triple(dilation)
        self.subm, self.inverse, self.transposed = subm, inverse, transposed
        self.indice_key = indice_key
        self.conv1x1 = all(k == 1 for k in self.kernel_size)
        self.weight = nn.Parameter(torch.empty(out_channels, *self.kernel_size, in_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter('bias', None)
        self._kio_cache = None
        self.reset_parameters()
        ks, st, pd = self.kernel_size, self.stride, self.padding
        ok = ((subm and ks == (3, 3, 3) and st == (1, 1, 1) and pd == (1, 1, 1))
              or (not subm and not inverse and ks == (2, 2, 2) and st == (2, 2, 2) and pd == (0, 0, 0))
              or (inverse and ks == (2, 2, 2)) or self.conv1x1)
        if not ok or any(d != 1 for d in self.dilation):
            raise NotImplementedError(
                f'softgroup_amd.spconv covers the conv shapes the SoftGroup model uses (SubM k3 p1, '
                f'SparseConv k2 s2 p0, SparseInverse k2, k1); got subm={subm} inverse={inverse} '
                f'kernel={ks} stride={st} padding={pd} dilation={self.dilation}')

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels * int(torch.tensor(self.kernel_size).prod())
            bound = 1 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        return (f'{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, '
                f'stride={self.stride}, subm={self.subm}, inverse={self.inverse}, '
                f'indice_key={self.indice_key}')

    def train(self, mode=True):
        self._kio_cache = None          # weights may change while training: re-pack afterwards
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):
        self._kio_cache = None
        return super()._apply(fn, *args, **kwargs)

    def _load_from_state_dict(self, *args, **kwargs):
        self._kio_cache = None
        return super()._load_from_state_dict(*args, **kwargs)

    def __getstate__(self):             # the packed copy is derived data: not pickled / deep-copied
        state = self.__dict__.copy()
        state['_kio_cache'] = None
        return state

    # [Cout, K, Cin] -> packed kernel layouts, cached until the weight changes: the forward layout and
    # (training) the transposed one of the input gradient, in fp32 and in bf16
    def weight_packed(self, transposed=False, bf16=False, flip_k=False):
        w = self.weight
        key = (_CACHE_EPOCH[0], w._version, w.data_ptr(), w.device, w.dtype)
        cache = self._kio_cache
        which = (transposed, bf16, flip_k)
        if cache is not None and cache[0] == key and which in cache[1]:
            return _acquired(cache[1][which])
        with _CACHE_FILL_LOCK:            # a miss: pack once, publish only when the data is there
            cache = self._kio_cache
            if cache is None or cache[0] != key:
                cache = (key, {})
            slot = cache[1]
            if which not in slot:
                kvol = int(torch.tensor(self.kernel_size).prod())
                cout, cin = self.out_channels, self.in_channels
                pack = pack_weight_bf16 if bf16 else pack_weight
                if not transposed:
                    packed = pack(w, cout, kvol, cin, False)
                else:
                    # w_t[k'][co][ci] = W[co][k][ci], k' = K-1-k for SubM (mirrored offset), k otherwise
                    w_t = w.detach().float().reshape(cout, kvol, cin).permute(1, 0, 2)
                    if flip_k:
                        w_t = w_t.flip(0)
                    packed = pack(w_t.contiguous(), cin, kvol, cout, True)
                slot[which] = (packed, _published(packed))
            self._kio_cache = cache
            return _acquired(slot[which])

    def _rule_and_plan(self, input):
        """-> (plan, out_indices, out_spatial_shape, backward-plan getter, mirrored offsets?)"""
        key = self.indice_key
        if self.subm:
            rule = input.find_indice_pair(key)
            if not isinstance(rule, SubMRule) or rule.num_rows != input.indices.shape[0]:
                rule = SubMRule(input.indices, input.spatial_shape)
                if key is not None:
                    input.indice_dict[key] = rule
            return rule.plan, input.indices, input.spatial_shape, (lambda: rule.plan), True
        if self.inverse:
            rule = input.find_indice_pair(key)
            if not isinstance(rule, DownRule):
                raise RuntimeError(f'SparseInverseConv3d: no SparseConv3d rulebook under indice_key '
                                   f'{key!r} (spconv requires the paired down conv to run first)')
            assert rule.num_out == input.indices.shape[0], 'inverse conv input does not match its pair'
            return rule.inv_plan, rule.in_indices, rule.in_spatial_shape, (lambda: rule.plan), False
        rule = input.find_indice_pair(key)
        if (not isinstance(rule, DownRule) or rule.num_in != input.indices.shape[0]
                or rule.in_indices.data_ptr() != input.indices.data_ptr()):
            rule = DownRule(input.indices, input.spatial_shape, input.batch_size)
            if key is not None:
                input.indice_dict[key] = rule
        return rule.plan, rule.out_indices, rule.out_spatial_shape, (lambda: rule.inv_plan), False

    def forward(self, input, post_scale=None, post_shift=None, residual=None):
        """post_scale/post_shift/residual: inference-only fused epilogue,
        out = relu((conv(x) + residual) * post_scale + post_shift)"""
        assert isinstance(input, SparseConvTensor)
        feats = input.features
        if self.conv1x1:   # plain GEMM on the active rows (blocks.py:31-41 semantics)
            assert post_scale is None and residual is None
            out = torch.mm(feats, self.weight.view(self.out_channels, self.in_channels).T)
            if self.bias is not None:
                out = out + self.bias
            return input.replace_feature(out)
        plan, out_indices, out_shape, bwd_plan, flip_k = self._rule_and_plan(input)
        # bf16 autocast (training, BASELINE config 3) or bf16 features: bf16 operands, fp32 sums;
        # everything else computes in fp32
        low = feats.is_cuda and (feats.dtype == torch.bfloat16 or (
            torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16))
        x = feats.to(torch.bfloat16 if low else torch.float32).contiguous()
        if torch.is_grad_enabled() and (feats.requires_grad or self.weight.requires_grad):
            assert post_scale is None and residual is None, 'fused epilogue is inference-only'
            packed = lambda t, b: self.weight_packed(t, b, flip_k and t)  # noqa: E731
            out = _GatherConvFn.apply(x, self.weight, packed, plan,
                                      bwd_plan() if x.requires_grad else None)
        elif low:
            assert post_scale is None and residual is None, 'fused epilogue is fp32-only'
            out = gather_conv_bf16(x, plan, self.weight_packed(False, True), self.out_channels)
        else:
            assert post_scale is None or self.bias is None
            out = gather_conv(x, plan, self.weight_packed(), self.out_channels, post_scale,
                              post_shift, residual)
        if self.bias is not None:
            out = out + self.bias.to(out.dtype)
        if not low and out.dtype != feats.dtype:
            out = out.to(feats.dtype)
        return SparseConvTensor(out, out_indices, out_shape, input.batch_size, input.grid,
                                input.voxel_num, input.indice_dict, input.benchmark)


class SubMConv3d(SparseConvolution):

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None, algo=None, fp32_accum=None, name=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                         bias, True, indice_key=indice_key, algo=algo, name=name)


class SparseConv3d(SparseConvolution):

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None, algo=None, fp32_accum=None, name=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                         bias, indice_key=indice_key, algo=algo, name=name)


class SparseInverseConv3d(SparseConvolution):

    def __init__(self, in_channels, out_channels, kernel_size, indice_key=None, bias=True, algo=None,
                 fp32_accum=None, name=None):
        super().__init__(3, in_channels, out_channels, kernel_size, bias=bias, inverse=True,
                         indice_key=indice_key, algo=algo, name=name)


# ------------------------------------------------------------------------------------------------
_CACHE_FILL_LOCK = threading.Lock()    # cache MISSES only: scans on several streams may share modules


class _Publication:
    """where and when a derived tensor was written: the producing stream and an event behind the
    write.  Consumers on another stream wait for the event ON THE DEVICE (once per stream); nobody
    blocks the host.  (Round 3 drained the producing stream instead, under the cache lock: with an
    unfrozen backbone every optimizer step invalidates every packed weight, i.e. two host-device
    drains per conv per step.)"""
    __slots__ = ('event', 'stream', 'seen')

    def __init__(self, device):
        st = torch.cuda.current_stream(device)
        self.event = torch.cuda.Event()
        self.event.record(st)
        self.stream = st.cuda_stream
        self.seen = {self.stream}


def _published(*tensors):
    """call right after enqueuing the writes of a freshly derived tensor on this thread's stream"""
    if tensors and tensors[0].is_cuda:
        return _Publication(tensors[0].device)
    return None


def _acquired(entry):
    """(tensor(s)..., publication) -> the tensor(s), ordered after their writes on this stream"""
    pub = entry[-1]
    if pub is not None:
        st = torch.cuda.current_stream()
        if st.cuda_stream not in pub.seen:
            st.wait_event(pub.event)
            for t in entry[:-1]:          # the allocator must not recycle it under this stream
                t.record_stream(st)
            pub.seen.add(st.cuda_stream)
    return entry[0] if len(entry) == 2 else entry[:-1]


def _bn_affine(bn):
    """eval-mode BatchNorm1d as y = x*scale + shift; cached until any of its tensors changes"""
    tensors = (bn.weight, bn.bias, bn.running_mean, bn.running_var)
    key = (_CACHE_EPOCH[0], ) + tuple((t._version, t.data_ptr()) if t is not None else None
                                      for t in tensors)
    cache = bn.__dict__.get('_sg_affine')
    if cache is None or cache[0] != key:
        with _CACHE_FILL_LOCK:
            cache = bn.__dict__.get('_sg_affine')
            if cache is None or cache[0] != key:
                with torch.no_grad():
                    inv = torch.rsqrt(bn.running_var.float() + bn.eps)
                    scale = inv * bn.weight.float() if bn.weight is not None else inv
                    shift = -bn.running_mean.float() * scale
                    if bn.bias is not None:
                        shift = shift + bn.bias.float()
                scale, shift = scale.contiguous(), shift.contiguous()
                cache = (key, scale, shift, _published(scale))
                bn.__dict__['_sg_affine'] = cache
    return _acquired(cache[1:])


def _needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _fusable_bn(m):
    return (isinstance(m, nn.BatchNorm1d) and not m.training and m.track_running_stats
            and m.running_mean is not None and not _needs_grad(m.weight, m.bias))


class SparseSequential(SparseModule):
    """spconv's sequential container: sparse modules receive the SparseConvTensor, dense modules
    (BatchNorm1d, ReLU, Identity ...) are applied to ``.features``.  Children are named
    '0','1',... or by the OrderedDict keys (state-dict compatibility).

    In eval mode BatchNorm1d -> ReLU runs as one elementwise kernel, and when it sits BETWEEN two
    sparse convs (blocks.py:57-70: BN, ReLU, conv, BN, ReLU, conv) it is the epilogue of the first
    conv; a residual is the epilogue of the last conv."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError('name exists.')
            self.add_module(name, module)

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError(f'index {idx} is out of range')
        if idx < 0:
            idx += len(self)
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
        self.add_module(name, module)

    def forward(self, input, residual=None):
        mods = list(self._modules.values())
        last_conv = max((i for i, m in enumerate(mods) if isinstance(m, SparseConvolution)),
                        default=-1)
        if residual is not None and last_conv != len(mods) - 1:
            raise ValueError('residual fusion needs the sequence to end in a sparse convolution')
        i = 0
        while i < len(mods):
            m = mods[i]
            sparse_in = isinstance(input, SparseConvTensor)
            fast = (sparse_in and input.features.is_cuda and input.features.dtype == torch.float32
                    and input.indices.shape[0] != 0 and not _needs_grad(input.features))
            if fast and _fusable_bn(m) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU):
                scale, shift = _bn_affine(m)
                input = input.replace_feature(bn_relu(input.features, scale, shift))
                i += 2
                continue
            if (fast and isinstance(m, SparseConvolution) and not m.conv1x1 and m.bias is None
                    and not _needs_grad(m.weight) and i + 2 < len(mods) and _fusable_bn(mods[i + 1])
                    and isinstance(mods[i + 2], nn.ReLU)):
                # conv -> BN -> ReLU -> (more): BN+ReLU is this conv's epilogue
                scale, shift = _bn_affine(mods[i + 1])
                input = m(input, post_scale=scale, post_shift=shift)
                i += 3
                continue
            if is_spconv_module(m):
                if isinstance(m, SparseConvolution) and i == last_conv and residual is not None:
                    input = m(input, residual=residual)
                else:
                    input = m(input)
            elif sparse_in:
                if input.indices.shape[0] != 0:
                    input = input.replace_feature(m(input.features))
            else:
                input = m(input)
            i += 1
        return input
