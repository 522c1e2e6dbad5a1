"""Building blocks of the SoftGroup backbone, hosted on softgroup_amd.spconv.

Module and parameter names are the checkpoint contract of the reference
(softgroup/model/blocks.py): ``MLP`` is an ``nn.Sequential`` (keys '0','1',..),
``ResidualBlock`` owns ``i_branch`` / ``conv_branch``, ``UBlock`` owns ``blocks`` / ``conv`` / ``u``
/ ``deconv`` / ``blocks_tail`` with children ``block0``, ``block1``.  Everything else -- how the
pieces are assembled and how they run -- is this package's own: inference goes through the fused
paths of ``SparseSequential`` (BatchNorm+ReLU as conv epilogues, residual add in the conv) or, for
whole U-Nets, through the native executor (spconv/unet_exec.py); the plain module composition
below is what training uses.
"""
from collections import OrderedDict

import torch
from torch import nn
import torch.nn.functional as F

from ..spconv import pytorch as spconv
from ..spconv.pytorch.modules import SparseModule


def _pre_activated(norm_fn, channels, conv):
    """the pre-activation triple every conv of the U-Net sits in: norm -> ReLU -> conv"""
    return [norm_fn(channels), nn.ReLU(), conv]


def _subm3(cin, cout, key):
    return spconv.SubMConv3d(cin, cout, kernel_size=3, padding=1, bias=False, indice_key=key)


class _PointLinearFn(torch.autograd.Function):
    """y = x @ w.T + b over ~1e5 point rows with <= 128 channels.  The weight gradient g.T @ x is a GEMM with a
    32 x 32 result and a reduction over all rows: the library GEMM runs it on ONE workgroup (269 us at 100 k rows,
    four of them per training step); here the rows are cut into <= 256 slabs (one batched GEMM, a workgroup each) whose
    results are added in slab order -- deterministic, ~15 us."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        gx = g @ w if ctx.needs_input_grad[0] else None
        gw = gb = None
        if ctx.needs_input_grad[1]:
            rows = x.shape[0]
            slabs = min(256, rows // 256)
            n = rows // slabs * slabs
            xs = x.detach()
            gw = torch.bmm(g[:n].view(slabs, n // slabs, -1).transpose(1, 2),
                           xs[:n].view(slabs, n // slabs, -1)).sum(0)
            if n < rows:
                gw = gw + g[n:].t() @ xs[n:]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g.sum(0)
        return gx, gw, gb


class PointLinear(nn.Linear):
    """nn.Linear for the point-wise heads (same parameters, same forward); its weight gradient is computed slab-wise
    when the input is a large fp32 CUDA matrix (see _PointLinearFn), otherwise it IS nn.Linear."""

    def forward(self, x):
        if (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[0] >= 16384 and x.is_contiguous()
                and torch.is_grad_enabled() and self.weight.requires_grad and not torch.is_autocast_enabled()
                and self.in_features <= 128 and self.out_features <= 128):
            return _PointLinearFn.apply(x, self.weight, self.bias)
        return F.linear(x, self.weight, self.bias)


class MLP(nn.Sequential):
    """num_layers-1 hidden layers of the input width (Linear, optional norm, ReLU) and a Linear to
    out_channels (reference blocks.py:9-27).  The heads start near zero: last layer N(0, 0.01)."""

    def __init__(self, in_channels, out_channels, norm_fn=None, num_layers=2):
        super().__init__()
        for _ in range(num_layers - 1):
            self.append(PointLinear(in_channels, in_channels))
            if norm_fn is not None:
                self.append(norm_fn(in_channels))
            self.append(nn.ReLU())
        self.append(PointLinear(in_channels, out_channels))

    def init_weights(self):
        for layer in self:
            if isinstance(layer, nn.Linear):
                nn.init.xavier_uniform_(layer.weight)
                nn.init.zeros_(layer.bias)
        head = self[len(self) - 1]
        nn.init.normal_(head.weight, 0, 0.01)
        nn.init.zeros_(head.bias)


class Custom1x1Subm3d(spconv.SparseConv3d):
    """1x1x1 "conv": a dense GEMM over the active rows, weight [Cout,1,1,1,Cin]
    (reference blocks.py:31-41).  The native executor runs it on the sparse-conv kernel with an
    identity gather table instead."""

    def forward(self, input):
        weight = self.weight.reshape(self.out_channels, self.in_channels)
        out = input.features @ weight.t()
        if self.bias is not None:
            out = out + self.bias
        return input.replace_feature(out)


class ResidualBlock(SparseModule):
    """x + SubM(ReLU(BN(SubM(ReLU(BN(x)))))) with a 1x1 projection on the identity branch when the
    width changes (reference blocks.py:44-79)."""

    def __init__(self, in_channels, out_channels, norm_fn, indice_key=None):
        super().__init__()
        same = in_channels == out_channels
        identity = nn.Identity() if same else Custom1x1Subm3d(in_channels, out_channels,
                                                              kernel_size=1, bias=False)
        self.i_branch = spconv.SparseSequential(identity)
        self.conv_branch = spconv.SparseSequential(
            *_pre_activated(norm_fn, in_channels, _subm3(in_channels, out_channels, indice_key)),
            *_pre_activated(norm_fn, out_channels, _subm3(out_channels, out_channels, indice_key)))

    def forward(self, input):
        shortcut = self.i_branch(input).features
        training_path = torch.is_grad_enabled() and (
            shortcut.requires_grad or any(p.requires_grad for p in self.conv_branch.parameters()))
        fusable = (not training_path and shortcut.is_cuda and shortcut.dtype == torch.float32
                   and input.indices.shape[0] != 0)
        if fusable:   # residual add happens in the epilogue of the second conv
            return self.conv_branch(input, residual=shortcut.contiguous())
        out = self.conv_branch(input)
        return out.replace_feature(out.features + shortcut)


def _named_blocks(block, widths, out_width, norm_fn, key):
    """block0, block1, ...: block i maps widths[i] -> out_width"""
    return spconv.SparseSequential(OrderedDict(
        (f'block{i}', block(w, out_width, norm_fn, indice_key=key)) for i, w in enumerate(widths)))


class UBlock(nn.Module):
    """One level of the sparse U-Net (reference blocks.py:82-143):
    blocks -> [strided conv -> inner UBlock -> inverse conv -> concat with the skip -> blocks_tail]."""

    def __init__(self, nPlanes, norm_fn, block_reps, block, indice_key_id=1):
        super().__init__()
        self.nPlanes = nPlanes
        width = nPlanes[0]
        subm_key, pair_key = f'subm{indice_key_id}', f'spconv{indice_key_id}'
        self.blocks = _named_blocks(block, [width] * block_reps, width, norm_fn, subm_key)
        if len(nPlanes) == 1:
            return
        inner = nPlanes[1]
        self.conv = spconv.SparseSequential(*_pre_activated(
            norm_fn, width, spconv.SparseConv3d(width, inner, kernel_size=2, stride=2, bias=False,
                                                indice_key=pair_key)))
        self.u = UBlock(nPlanes[1:], norm_fn, block_reps, block, indice_key_id=indice_key_id + 1)
        self.deconv = spconv.SparseSequential(*_pre_activated(
            norm_fn, inner, spconv.SparseInverseConv3d(inner, width, kernel_size=2, bias=False,
                                                       indice_key=pair_key)))
        # the first tail block sees [skip | upsampled]: twice the width
        self.blocks_tail = _named_blocks(block, [2 * width] + [width] * (block_reps - 1), width,
                                         norm_fn, subm_key)

    def forward(self, input):
        x = self.blocks(input)
        if len(self.nPlanes) == 1:
            return x
        up = self.deconv(self.u(self.conv(x)))
        merged = x.replace_feature(torch.cat((x.features, up.features), dim=1))
        return self.blocks_tail(merged)
