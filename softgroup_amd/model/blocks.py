"""Building blocks of the SoftGroup backbone, hosted on softgroup_amd.spconv.

Module/parameter names follow the reference (softgroup/model/blocks.py) exactly, because they
are the checkpoint contract: ``MLP`` is an ``nn.Sequential`` (keys '0','1',..),
``ResidualBlock`` owns ``i_branch`` / ``conv_branch``, ``UBlock`` owns ``blocks`` / ``conv`` /
``u`` / ``deconv`` / ``blocks_tail`` with children ``block0``, ``block1``.
"""
from collections import OrderedDict

import torch
from torch import nn

from ..spconv import pytorch as spconv
from ..spconv.pytorch.modules import SparseModule


class MLP(nn.Sequential):
    """(Linear -> [norm] -> ReLU) x (num_layers-1) -> Linear   (reference blocks.py:9-27)"""

    def __init__(self, in_channels, out_channels, norm_fn=None, num_layers=2):
        layers = []
        for _ in range(num_layers - 1):
            layers.append(nn.Linear(in_channels, in_channels))
            if norm_fn:
                layers.append(norm_fn(in_channels))
            layers.append(nn.ReLU())
        layers.append(nn.Linear(in_channels, out_channels))
        super().__init__(*layers)

    def init_weights(self):
        linears = [m for m in self.modules() if isinstance(m, nn.Linear)]
        for lin in linears:
            nn.init.xavier_uniform_(lin.weight)
            nn.init.constant_(lin.bias, 0)
        nn.init.normal_(self[-1].weight, 0, 0.01)
        nn.init.constant_(self[-1].bias, 0)


class Custom1x1Subm3d(spconv.SparseConv3d):
    """1x1x1 "conv" = plain GEMM over the active rows; weight [Cout,1,1,1,Cin]
    (reference blocks.py:31-41)."""

    def forward(self, input):
        w = self.weight.view(self.out_channels, self.in_channels)
        feats = torch.mm(input.features, w.T)
        if self.bias is not None:
            feats = feats + self.bias
        return input.replace_feature(feats)


class ResidualBlock(SparseModule):
    """pre-activation residual block: x + SubM(ReLU(BN(SubM(ReLU(BN(x))))))
    (reference blocks.py:44-79).  In eval mode both BN+ReLU pairs and the residual add are fused
    into the two conv kernels by SparseSequential."""

    def __init__(self, in_channels, out_channels, norm_fn, indice_key=None):
        super().__init__()
        if in_channels == out_channels:
            self.i_branch = spconv.SparseSequential(nn.Identity())
        else:
            self.i_branch = spconv.SparseSequential(
                Custom1x1Subm3d(in_channels, out_channels, kernel_size=1, bias=False))
        self.conv_branch = spconv.SparseSequential(
            norm_fn(in_channels), nn.ReLU(),
            spconv.SubMConv3d(in_channels, out_channels, kernel_size=3, padding=1, bias=False,
                              indice_key=indice_key),
            norm_fn(out_channels), nn.ReLU(),
            spconv.SubMConv3d(out_channels, out_channels, kernel_size=3, padding=1, bias=False,
                              indice_key=indice_key))

    def forward(self, input):
        shortcut = self.i_branch(input).features
        needs_grad = torch.is_grad_enabled() and (
            shortcut.requires_grad or any(p.requires_grad for p in self.conv_branch.parameters()))
        fuse = (not needs_grad and shortcut.is_cuda and shortcut.dtype == torch.float32
                and input.indices.shape[0] != 0)
        if fuse:
            return self.conv_branch(input, residual=shortcut.contiguous())
        out = self.conv_branch(input)
        return out.replace_feature(out.features + shortcut)


class UBlock(nn.Module):
    """One level of the sparse U-Net (reference blocks.py:82-143):
    blocks -> [down conv -> inner UBlock -> inverse conv -> concat skip -> blocks_tail]."""

    def __init__(self, nPlanes, norm_fn, block_reps, block, indice_key_id=1):
        super().__init__()
        self.nPlanes = nPlanes
        c = nPlanes[0]
        subm_key = f'subm{indice_key_id}'
        self.blocks = spconv.SparseSequential(OrderedDict(
            (f'block{i}', block(c, c, norm_fn, indice_key=subm_key)) for i in range(block_reps)))
        if len(nPlanes) > 1:
            down_key = f'spconv{indice_key_id}'
            self.conv = spconv.SparseSequential(
                norm_fn(c), nn.ReLU(),
                spconv.SparseConv3d(c, nPlanes[1], kernel_size=2, stride=2, bias=False,
                                    indice_key=down_key))
            self.u = UBlock(nPlanes[1:], norm_fn, block_reps, block, indice_key_id=indice_key_id + 1)
            self.deconv = spconv.SparseSequential(
                norm_fn(nPlanes[1]), nn.ReLU(),
                spconv.SparseInverseConv3d(nPlanes[1], c, kernel_size=2, bias=False,
                                           indice_key=down_key))
            self.blocks_tail = spconv.SparseSequential(OrderedDict(
                (f'block{i}', block(c * (2 - i), c, norm_fn, indice_key=subm_key))
                for i in range(block_reps)))

    def forward(self, input):
        x = self.blocks(input)
        if len(self.nPlanes) == 1:
            return x
        skip = x.features
        y = self.deconv(self.u(self.conv(x)))
        x = x.replace_feature(torch.cat((skip, y.features), dim=1))
        return self.blocks_tail(x)
