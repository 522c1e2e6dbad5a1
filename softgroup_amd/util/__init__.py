from .cast import cuda_cast, force_fp32, to_host, to_host_begin, to_host_end  # noqa: F401
from .rle import (rle_decode, rle_encode, rle_encode_many, rle_encode_runs,  # noqa: F401
                  rle_text_to_dicts)  # noqa: F401
