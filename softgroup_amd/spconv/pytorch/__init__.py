from ..core import (invalidate_caches, SparseConv3d, SparseConvTensor, SparseConvolution, SparseInverseConv3d,  # noqa: F401
                    SparseModule, SparseSequential, SubMConv3d, is_spconv_module)
from . import modules  # noqa: F401
