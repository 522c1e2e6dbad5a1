from ..core import SparseModule, SparseSequential, is_spconv_module  # noqa: F401
