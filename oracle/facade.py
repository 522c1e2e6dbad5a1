"""TEST INFRASTRUCTURE: runs the REFERENCE'S OWN Python model (/root/reference/softgroup/model/
softgroup.py + blocks.py, imported from where it lies, nothing copied) on the CPU, on top of
CPU stand-ins for its two native dependencies built from the C oracle:

  * ``spconv.pytorch``  -> the classes below (SparseConvTensor, SubMConv3d, SparseConv3d,
    SparseInverseConv3d, SparseSequential, modules.SparseModule) with spconv-2 weight layout
    [Cout,k,k,k,Cin] and the semantics of SURVEY 2.4 (oracle/sg_oracle_conv.c, itself pinned
    against dense F.conv3d / F.conv_transpose3d);
  * ``softgroup.ops``   -> the functions of ``OPS`` below with the signatures of the reference's
    softgroup/ops/functions.py, on CPU tensors (oracle/sg_oracle.c, pinned against the reference's
    own CPU ops and -- on the GPU box -- against its own CUDA kernels, tests/test_ref_gpu_kernels.py).

What this buys: the reference's control flow (forward_test, the per-class grouping loop with
get_batch_offsets, clusters_voxelization, forward_instance / global_pool, get_instances with dense
masks and rle_encode) is executed AS WRITTEN, so its outputs can be stored as golden vectors
(tests/golden/make_ref_forward.py -> tests/golden/ref_forward_*.npz) that pin both oracle/model.py
(the restatement of that control flow) and the HIP-hosted model.  The reference hard-codes
``.cuda()`` / ``device='cuda'`` in a few places; ``reference_model()`` neutralises exactly those
(Tensor.cuda -> identity, torch.zeros/tensor/ones(device='cuda') -> CPU) while the model runs.

Only tests/ and tests/golden/*.py import this module.  The reference's Python is imported from
/root/reference where that exists (authoring container) and otherwise from oracle/_ref/pysrc --
the same files byte-compiled by oracle/build_ref.py (sourceless .pyc; a git-ignored build output
that travels to the GPU box like the .so files)."""
import contextlib
import importlib
import importlib.machinery
import importlib.util
import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch
from torch import nn

import oracle as O

REF_ROOT = os.environ.get('SG_REF_ROOT', '/root/reference')   # (override: exercise the .pyc path here)
PYC_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ref', 'pysrc')


def ref_root():
    """directory holding the reference's `softgroup/` and `tools/` Python: the reference tree itself,
    or its byte-compiled image under oracle/_ref/pysrc (None if neither exists)"""
    if os.path.isdir(os.path.join(REF_ROOT, 'softgroup', 'model')):
        return REF_ROOT
    if os.path.isdir(os.path.join(PYC_ROOT, 'softgroup', 'model')):
        magic = open(os.path.join(PYC_ROOT, 'MAGIC'), 'rb').read()
        assert magic == importlib.util.MAGIC_NUMBER, 'oracle/_ref/pysrc was compiled by another Python'
        return PYC_ROOT
    return None


def _t(a, dtype=None):
    x = torch.from_numpy(np.ascontiguousarray(a))
    return x if dtype is None else x.to(dtype)


def _n(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------------------------ spconv.pytorch
class SparseConvTensor:

    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        self.features = features
        self.indices = indices
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = batch_size
        self.indice_dict = {}
        self.grid = grid

    def replace_feature(self, feature):
        out = SparseConvTensor(feature, self.indices, self.spatial_shape, self.batch_size, self.grid)
        out.indice_dict = self.indice_dict
        return out


class SparseModule(nn.Module):
    pass


class SparseSequential(SparseModule):

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for k, m in args[0].items():
                self.add_module(k, m)
        else:
            for i, m in enumerate(args):
                self.add_module(str(i), m)
        for k, m in kwargs.items():
            self.add_module(k, m)

    def forward(self, input):
        for m in self._modules.values():
            if isinstance(m, SparseModule):
                input = m(input)
            elif isinstance(input, SparseConvTensor):
                if input.indices.shape[0] != 0:
                    input = input.replace_feature(m(input.features))
            else:
                input = m(input)
        return input


class _Conv(SparseModule):

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None, **kw):
        super().__init__()
        assert not bias and groups == 1 and dilation == 1
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.indice_key = indice_key
        k = kernel_size
        self.weight = nn.Parameter(torch.zeros(out_channels, k, k, k, in_channels))
        self.bias = None


class _GatherConvFn(torch.autograd.Function):
    """out[j] = sum_k W[:, k, :] . in[table[j, k]] (table entry -1: no term) -- the one operator behind
    the three conv flavours (oracle/sg_oracle_conv.c `gather_conv`).  Forward: the C oracle itself, so
    values are those of the non-differentiable stand-ins.  Backward: the textbook gradients of that
    sum, accumulated in float64 -- d in[i] += W[:, k, :]^T . d out[j] over the pairs (j, k) that
    gathered row i, d W[:, k, :] = sum_j d out[j] (x) in[table[j, k]] -- i.e. what spconv's backward
    computes (its source is not vendored), used by tests/golden/make_ref_train.py for the
    GRADIENT goldens of the reference's forward_train."""

    @staticmethod
    def forward(ctx, feats, weight, table, fwd):
        ctx.save_for_backward(feats, weight)
        ctx.table = table
        return _t(fwd(_n(feats), _n(weight)))

    @staticmethod
    def backward(ctx, g):
        feats, weight = ctx.saved_tensors
        tab = torch.from_numpy(np.ascontiguousarray(ctx.table).astype(np.int64))
        K, cout, cin = tab.shape[1], weight.shape[0], feats.shape[1]
        W = weight.detach().reshape(cout, K, cin).double()
        x, g = feats.detach().double(), g.double()
        gx, gW = torch.zeros_like(x), torch.zeros_like(W)
        for k in range(K):
            m = tab[:, k] >= 0
            if not bool(m.any()):
                continue
            idx, gk = tab[m, k], g[m]
            gW[:, k, :] = gk.t() @ x[idx]
            gx.index_add_(0, idx, gk @ W[:, k, :])
        return gx.to(feats.dtype), gW.reshape(weight.shape).to(weight.dtype), None, None


def _conv(feats, weight, table, fwd):
    if torch.is_grad_enabled() and (feats.requires_grad or weight.requires_grad):
        return _GatherConvFn.apply(feats, weight, table, fwd)
    return _t(fwd(_n(feats), _n(weight)))


class SubMConv3d(_Conv):

    def forward(self, x):
        assert self.kernel_size == 3 and self.padding == 1
        if self.indice_key not in x.indice_dict:
            x.indice_dict[self.indice_key] = O.subm_rulebook(_n(x.indices), x.spatial_shape)
        nbr = x.indice_dict[self.indice_key]
        return x.replace_feature(_conv(x.features, self.weight, nbr, lambda f, w: O.subm_conv3d(f, nbr, w)))


class SparseConv3d(_Conv):

    def forward(self, x):
        assert self.kernel_size == 2 and self.stride == 2 and self.padding == 0
        out_idx, in2out, child, oshape = O.down_rulebook(_n(x.indices), x.spatial_shape)
        out = _conv(x.features, self.weight, child, lambda f, w: O.sparse_conv3d_k2s2(f, child, w))
        y = SparseConvTensor(out, _t(out_idx), oshape, x.batch_size, x.grid)
        y.indice_dict = x.indice_dict
        y.indice_dict[self.indice_key] = (x.indices, in2out, x.spatial_shape)
        return y


class SparseInverseConv3d(_Conv):

    def forward(self, x):
        fine_indices, in2out, fine_shape = x.indice_dict[self.indice_key]
        fi = _n(fine_indices)
        kk = (fi[:, 1] & 1) * 4 + (fi[:, 2] & 1) * 2 + (fi[:, 3] & 1)
        table = np.full((fi.shape[0], 8), -1, np.int32)            # sg_oracle_conv.c:156-166
        table[np.arange(fi.shape[0]), kk] = in2out
        out = _conv(x.features, self.weight, table, lambda f, w: O.inverse_conv3d_k2(f, fi, in2out, w))
        y = SparseConvTensor(out, fine_indices, fine_shape, x.batch_size, x.grid)
        y.indice_dict = x.indice_dict
        return y


class _VoxelizationFn(torch.autograd.Function):
    """softgroup/ops/functions.py:105-150 (`Voxelization`): forward voxelize_fp, backward voxelize_bp"""

    @staticmethod
    def forward(ctx, feats, map_rule, mode):
        ctx.map_rule, ctx.mode, ctx.n = _n(map_rule), mode, feats.shape[0]
        return _t(O.voxelization(_n(feats), ctx.map_rule, mode))

    @staticmethod
    def backward(ctx, g):
        return _t(O.voxelization_bp(_n(g.contiguous()), ctx.map_rule, ctx.n, ctx.mode)), None, None


class _GlobalAvgPoolFn(torch.autograd.Function):
    """softgroup/ops/functions.py:333-371 (`GlobalAvgPool`): forward / backward of the ROI average pool"""

    @staticmethod
    def forward(ctx, feats, offsets):
        ctx.offsets, ctx.n = _n(offsets), feats.shape[0]
        return _t(O.global_avg_pool(_n(feats), ctx.offsets))

    @staticmethod
    def backward(ctx, g):
        return _t(O.global_avg_pool_bp(_n(g.contiguous()), ctx.offsets, ctx.n)), None


def _spconv_modules():
    top = types.ModuleType('spconv')
    pt = types.ModuleType('spconv.pytorch')
    mods = types.ModuleType('spconv.pytorch.modules')
    for name in ('SparseConvTensor', 'SparseModule', 'SparseSequential', 'SubMConv3d', 'SparseConv3d',
                 'SparseInverseConv3d'):
        setattr(pt, name, globals()[name])
    mods.SparseModule = SparseModule
    pt.modules = mods
    top.pytorch = pt
    return {'spconv': top, 'spconv.pytorch': pt, 'spconv.pytorch.modules': mods}


# ------------------------------------------------------------------------------------ softgroup.ops
def _ops_module():
    m = types.ModuleType('softgroup.ops')

    def voxelization_idx(coords, batchsize, mode=4):
        oc, im, om = O.voxelization_idx(_n(coords), batchsize, mode)
        return _t(oc), _t(im), _t(om)

    def voxelization(feats, map_rule, mode=4):
        if torch.is_grad_enabled() and feats.requires_grad:
            return _VoxelizationFn.apply(feats, map_rule, mode)
        return _t(O.voxelization(_n(feats), _n(map_rule), mode))

    def global_avg_pool(feats, offsets):
        if torch.is_grad_enabled() and feats.requires_grad:
            return _GlobalAvgPoolFn.apply(feats, offsets)
        return _t(O.global_avg_pool(_n(feats), _n(offsets)))

    def ballquery_batch_p(coords, batch_idxs, batch_offsets, radius, meanActive):
        idx, sl = O.ballquery_batch_p(_n(coords), _n(batch_idxs), _n(batch_offsets), radius, meanActive)
        return _t(idx), _t(sl)

    def octree_ball_query(coords, mean_active, radius):
        idx, sl = O.octree_ball_query(_n(coords), mean_active, radius)
        return _t(idx), _t(sl)

    def ball_query(coords, batch_idxs, batch_offsets, radius, mean_active, with_octree=False):
        if with_octree:
            return octree_ball_query(coords, mean_active, radius)
        return ballquery_batch_p(coords, batch_idxs, batch_offsets, radius, mean_active)

    def bfs_cluster(class_numpoint_mean, ball_query_idxs, start_len, threshold, class_id):
        ci, co = O.bfs_cluster(_n(class_numpoint_mean), _n(ball_query_idxs), _n(start_len), threshold,
                               class_id)
        return _t(ci), _t(co)

    def _seg(name):
        return lambda inp, offsets: _t(getattr(O, name)(_n(inp), _n(offsets)))

    def get_mask_iou_on_cluster(proposals_idx, proposals_offset, instance_labels, instance_pointnum):
        return _t(O.get_mask_iou_on_cluster(_n(proposals_idx), _n(proposals_offset),
                                            _n(instance_labels), _n(instance_pointnum)))

    def get_mask_iou_on_pred(proposals_idx, proposals_offset, instance_labels, instance_pointnum,
                             mask_scores_sigmoid):
        return _t(O.get_mask_iou_on_pred(_n(proposals_idx), _n(proposals_offset), _n(instance_labels),
                                         _n(instance_pointnum), _n(mask_scores_sigmoid)))

    def get_mask_label(proposals_idx, proposals_offset, instance_labels, instance_cls,
                       instance_pointnum, proposals_iou, iou_thr):
        return _t(O.get_mask_label(_n(proposals_idx), _n(proposals_offset), _n(instance_labels),
                                   _n(instance_cls), _n(instance_pointnum), _n(proposals_iou), iou_thr))

    for k, v in dict(voxelization_idx=voxelization_idx, voxelization=voxelization,
                     ballquery_batch_p=ballquery_batch_p, octree_ball_query=octree_ball_query,
                     ball_query=ball_query, bfs_cluster=bfs_cluster, global_avg_pool=global_avg_pool,
                     sec_min=_seg('sec_min'), sec_max=_seg('sec_max'), sec_mean=_seg('sec_mean'),
                     get_mask_iou_on_cluster=get_mask_iou_on_cluster,
                     get_mask_iou_on_pred=get_mask_iou_on_pred, get_mask_label=get_mask_label).items():
        setattr(m, k, v)
    return m


# ------------------------------------------------------------------------------------ the reference model
class _TorchOnCpu:
    """`torch` as seen by the reference model module: factory calls that hard-code device='cuda'
    (softgroup.py:156,568,669) land on the CPU; everything else is torch itself."""

    def __init__(self):
        self._torch = torch

    def __getattr__(self, name):
        attr = getattr(self._torch, name)
        if name in ('zeros', 'ones', 'tensor', 'full', 'empty', 'rand'):
            def on_cpu(*a, **k):
                if str(k.get('device', 'cpu')).startswith('cuda'):
                    k['device'] = 'cpu'
                return attr(*a, **k)
            return on_cpu
        return attr


class _ScalarLog:
    """stand-in for tensorboardX.SummaryWriter (softgroup/util/logger.py:26-38 subclasses it):
    keeps every (tag, value, step) of add_scalar in `.scalars`"""
    last = None

    def __init__(self, *args, **kwargs):
        self.scalars = []
        _ScalarLog.last = self

    def add_scalar(self, tag, value, step=None, *args, **kwargs):
        self.scalars.append((tag, float(value), step))

    def flush(self, *args, **kwargs):
        pass


class NS(dict):
    """config section with attribute access (the reference uses munch.Munch, not installed here)"""
    __getattr__ = dict.get


def import_reference(ops_module=None, spconv_modules=None):
    """Import /root/reference/softgroup/model with the given stand-ins for `softgroup.ops` and
    `spconv.pytorch` (default: the oracle-backed CPU ones of this file).  Returns the module
    `softgroup.model.softgroup`."""
    root = ref_root()
    assert root is not None, 'needs /root/reference or oracle/_ref/pysrc (oracle/build_ref.py)'
    for k in [k for k in sys.modules if k == 'softgroup' or k.startswith('softgroup.')]:
        del sys.modules[k]
    sys.modules.update(spconv_modules or _spconv_modules())
    tb = types.ModuleType('tensorboardX')
    tb.SummaryWriter = _ScalarLog          # tensorboardX is not installed: tools/train.py's writer records here
    sys.modules.setdefault('tensorboardX', tb)
    ply = types.ModuleType('plyfile')          # instance_eval_util imports it for file export only
    ply.PlyData = ply.PlyElement = object
    sys.modules.setdefault('plyfile', ply)
    pkg = types.ModuleType('softgroup')
    pkg.__path__ = [os.path.join(root, 'softgroup')]
    sys.modules['softgroup'] = pkg
    sys.modules['softgroup.ops'] = ops_module or _ops_module()
    return importlib.import_module('softgroup.model.softgroup')


class Munch(dict):
    """stand-in for munch.Munch (not installed): dict with attribute access, as tools/test.py and
    tools/train.py use it (`Munch.fromDict(yaml)`, `cfg.model`, `getattr(cfg, key, default)`)"""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = dict.__setitem__

    @classmethod
    def fromDict(cls, d):
        if isinstance(d, dict):
            return cls((k, cls.fromDict(v)) for k, v in d.items())
        if isinstance(d, (list, tuple)):
            return type(d)(cls.fromDict(v) for v in d)
        return d


def torch2_compat(ref_module):
    """The reference was written for torch 1.11 (docs/installation.md:14-19).  One construct of its
    `get_instances` no longer runs on torch >= 2: `proposals_idx[mask_inds]` indexes the CPU tensor
    that `bfs_cluster` returns with a CUDA boolean mask (softgroup.py:568-569); torch 2 raises
    "indices should be either on cpu or on the same device as the indexed tensor".  This moves that
    one argument to the scores' device before the reference's own method body runs -- no logic of
    the reference is replaced.  (INTEGRATION.md section 3 lists it as the one-line change a
    maintainer on torch 2 needs regardless of the operator library underneath.)"""
    cls = ref_module.SoftGroup
    if getattr(cls, '_sg_torch2_compat', False):
        return
    orig = cls.get_instances

    def get_instances(self, scan_id, proposals_idx, semantic_scores, *a, **k):
        return orig(self, scan_id, proposals_idx.to(semantic_scores.device), semantic_scores, *a, **k)

    cls.get_instances = get_instances
    cls._sg_torch2_compat = True


def import_reference_tool(name):
    """the reference's tools/<name>.py as a module (call after import_reference, which installs the
    `softgroup` package the tool imports)"""
    root = ref_root()
    m = types.ModuleType('munch')
    m.Munch = Munch
    sys.modules.setdefault('munch', m)
    src = os.path.join(root, 'tools', name + '.py')
    path = src if os.path.exists(src) else src + 'c'
    loader = (importlib.machinery.SourceFileLoader if path.endswith('.py')
              else importlib.machinery.SourcelessFileLoader)(f'ref_tools_{name}', path)
    spec = importlib.util.spec_from_loader(loader.name, loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod


@contextlib.contextmanager
def cpu_only(ref_module):
    """while active: Tensor.cuda() is the identity and the reference module's device='cuda'
    factories produce CPU tensors"""
    saved_cuda, saved_torch = torch.Tensor.cuda, ref_module.torch
    torch.Tensor.cuda = lambda self, *a, **k: self
    ref_module.torch = _TorchOnCpu()
    try:
        yield
    finally:
        torch.Tensor.cuda = saved_cuda
        ref_module.torch = saved_torch


def reference_model(cfg, state_dict):
    """the reference's SoftGroup(**cfg) on the oracle-backed CPU stand-ins, weights from
    ``state_dict`` (reference key names); returns (model, module) -- run it inside cpu_only(module)"""
    mod = import_reference()
    c = dict(cfg)
    for k in ('grouping_cfg', 'instance_voxel_cfg', 'train_cfg', 'test_cfg'):
        if c.get(k) is not None:
            c[k] = NS(c[k])
    model = mod.SoftGroup(**c)
    missing, unexpected = model.load_state_dict({k: v.detach().cpu() for k, v in state_dict.items()},
                                                strict=True)
    assert not missing and not unexpected
    model.eval()      # (the reference's train() override returns None, softgroup.py:102-110)
    return model, mod
