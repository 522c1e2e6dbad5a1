"""Drop-in for the subset of ``spconv`` (2.1.x) that the reference SoftGroup model imports.

``import softgroup_amd.spconv.pytorch as spconv`` mirrors ``import spconv.pytorch as spconv``
(reference: softgroup/model/softgroup.py:5, blocks.py:3-5, util/fp16.py:8)."""
from .core import (invalidate_caches, SparseConv3d, SparseConvTensor, SparseConvolution, SparseInverseConv3d,  # noqa: F401
                   SparseModule, SparseSequential, SubMConv3d, gather_conv, is_spconv_module)
