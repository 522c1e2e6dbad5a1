"""ctypes binding of libsoftgroup_hip.so (the C ABI in include/softgroup_hip.h).

The library is the product: there is NO fallback.  If it is missing, importing any op raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', os.environ.get('SG_LIB_NAME', 'libsoftgroup_hip.so'))   # (developer A/B builds)

_lib = None

_vp, _i, _f, _i64, _sz = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_size_t
_pi32 = C.POINTER(C.c_int32)

# name -> (restype, argtypes).  Must list every symbol of include/softgroup_hip.h
# (tests/test_cabi_symbols.py checks the two against each other).
SIGNATURES = {
    'sg_version': (_i, []),
    'sg_last_error': (C.c_char_p, []),
    'sg_device_info': (_i, [C.c_char_p, _i, C.POINTER(_i), C.POINTER(_i)]),
    'sg_stream_release': (_i, [_vp]),
    'sg_host_copy_2d': (_i, [_vp, _i64, _vp, _i64, _i64, _i64]),
    'sg_host_cast_f64_f32': (_i, [_vp, _vp, _i64]),
    'sg_host_fill_i64_strided': (_i, [_vp, _i64, _i64, _i64]),
    'sg_host_colmax_i64': (_i, [_vp, _i64, _i64, _vp]),
    'sg_stream_create': (_i, [_vp]),
    'sg_stream_create_priority': (_i, [_vp, _i]),
    'sg_stream_destroy': (_i, [_vp]),
    'sg_voxelize_idx_host': (_i, [_vp, _i, _i, _i, _vp, _pi32, _pi32]),
    'sg_voxelize_idx_fill_host': (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp, _vp]),
    'sg_voxelize_idx_workspace_bytes': (_sz, [_i]),
    'sg_voxelize_idx_build': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    'sg_voxelize_idx_fill': (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    'sg_voxelize_fp': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'sg_voxelize_bp': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'sg_ballquery_workspace_bytes': (_sz, [_i]),
    'sg_ballquery_build_grid': (_i, [_vp, _vp, _i, _f, _vp, _sz, _vp]),
    'sg_ballquery_count': (_i, [_vp, _vp, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    'sg_ballquery_fill': (_i, [_vp, _vp, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    'sg_scan_workspace_bytes': (_sz, [_i]),
    'sg_exclusive_scan_startlen': (_i, [_vp, _i, _vp, _vp, _sz, _vp]),
    'sg_octree_build_host': (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    'sg_octree_build_workspace_bytes': (_sz, [_i]),
    'sg_octree_build': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    'sg_pyramid_inverse_map_workspace_bytes': (_sz, [_i, _i, _i]),
    'sg_pyramid_inverse_map': (_i, [_vp, _i64, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    'sg_octree_ballquery_count': (_i, [_vp, _vp, _vp, _vp, _i, _f, _vp, _vp]),
    'sg_octree_ballquery_fill': (_i, [_vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp]),
    'sg_bfs_workspace_bytes': (_sz, [_i, _i64]),
    'sg_bfs_cluster_label': (_i, [_vp, _vp, _i, _i64, _i, _vp, _vp, _i, _pi32, _pi32, _vp, _sz, _vp]),
    'sg_bfs_cluster_emit': (_i, [_vp, _vp, _i, _i64, _vp, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    'sg_sec_mean': (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    'sg_sec_min': (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    'sg_sec_max': (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    'sg_global_avg_pool_fp': (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    'sg_global_avg_pool_bp': (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    'sg_get_mask_iou_on_cluster': (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    'sg_get_mask_iou_on_pred': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    'sg_get_mask_label': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp]),
    'sg_spconv_hash_workspace_bytes': (_sz, [_i]),
    'sg_spconv_subm_rulebook': (_i, [_vp, _i, _pi32, _vp, _vp, _sz, _vp]),
    'sg_spconv_pyramid_workspace_bytes': (_sz, [_i, _i]),
    'sg_spconv_pyramid_rows': (_i, [_vp, _i, _pi32, _i, _vp, _vp, _sz, _vp]),
    'sg_spconv_pyramid_build_workspace_bytes': (_sz, [_vp, _i]),
    'sg_spconv_pyramid_build': (_i, [_vp, _i, _pi32, _i, _vp, _vp, _sz, _vp, _sz, _vp]),
    'sg_spconv_down_build': (_i, [_vp, _i, _pi32, _vp, _vp, _vp, _sz, _vp]),
    'sg_spconv_down_fill': (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    'sg_spconv_inverse_rulebook': (_i, [_vp, _vp, _i, _vp, _vp]),
    'sg_spconv_plan_workspace_bytes': (_sz, [_i]),
    'sg_spconv_plan': (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    'sg_spconv_set_arithmetic': (_i, [_i]),
    'sg_spconv_set_combine': (_i, [_i]),
    'sg_spconv_set_chain': (_i, [_i]),
    'sg_spconv_chain_stats': (_i, [_vp, _vp]),
    'sg_spconv_profile': (_i, [_i]),
    'sg_spconv_profile_read': (_i, [_vp, _vp]),
    'sg_spconv_profile_detail': (_i, [_vp, _vp, _i, _vp]),
    'sg_unet_arena_bytes': (_sz, [_vp, _i]),
    'sg_unet_forward': (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    'sg_unet_train_arena_hint': (_sz, [_vp, _i]),
    'sg_unet_train_forward': (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    'sg_unet_train_backward': (_i, [_vp, _vp, _vp, _vp]),
    'sg_unet_train_release': (None, [_vp]),
    'sg_pointwise_heads': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'sg_spconv_packed_weight_elems': (_sz, [_i, _i, _i]),
    'sg_spconv_pack_weight': (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    'sg_spconv_conv_workspace_bytes': (_sz, [_i, _i]),
    'sg_spconv_gather_conv_f32': (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                       _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'sg_spconv_packed_weight_elems_bf16': (_sz, [_i, _i, _i]),
    'sg_spconv_pack_weight_bf16': (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    'sg_spconv_conv_bf16_workspace_bytes': (_sz, [_i, _i]),
    'sg_spconv_gather_conv_bf16': (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                        _vp]),
    'sg_spconv_transpose_table': (_i, [_vp, _i, _i, _vp, _vp]),
    'sg_spconv_wgrad_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'sg_spconv_wgrad': (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    'sg_instance_npoint': (_i, [_vp, _vp, _i64, _i, _i, _f, _i, _vp, _vp]),
    'sg_instance_runs_workspace_bytes': (_sz, [_i, _i]),
    'sg_instance_runs': (_i, [_vp, _vp, _i64, _i, _i, _f, _vp, _i, _i, _i, _vp, _vp, _vp, _i64, _vp, _sz, _vp]),
    'sg_rle_format_bound': (_i64, [_i64, _i]),
    'sg_rle_format_host': (_i, [_vp, _vp, _vp, _i, _vp, _i64, _vp]),
    'sg_rle_format_device_workspace_bytes': (_sz, [_i64]),
    'sg_rle_format_device_text_bytes': (_i64, [_i64, _i64]),
    'sg_rle_format_device': (_i, [_vp, _vp, _vp, _i, _i64, _i64, _vp, _i64, _vp, _vp, _sz, _vp]),
    'sg_rle_format_runs_host': (_i, [_vp, _vp, _vp, _i, _vp, _i64, _vp]),
    'sg_panoptic_fusion_workspace_bytes': (_sz, [_i, _i]),
    'sg_panoptic_fusion': (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, C.c_double, _i, _i, _vp, _vp, _sz, _vp]),
    'sg_scan_grouping': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    'sg_scan_grouping_pp': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    'sg_scan_instances': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp, _vp]),
    'sg_softmax_rows': (_i, [_vp, _i64, _i, _vp, _vp]),
    'sg_mlp_rows': (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp]),
    'sg_linear_rows': (_i, [_vp, _i64, _vp, _vp, _vp]),
    'sg_scan_arena_bytes': (_sz, [_vp, _i, _i]),
    'sg_scan_forward': (_i, [_vp, _vp, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _vp]),
    'sg_eval_intersections': (_i, [_vp, _vp, _vp, _i, _i64, _vp, _i, _i, _vp, _vp]),
    'sg_bn_relu_f32': (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp, _vp]),
    'sg_gather_rows_f32': (_i, [_vp, _vp, _i64, _i, _vp, _vp]),
    'sg_gather_rows_i64idx_f32': (_i, [_vp, _vp, _i64, _i, _vp, _vp]),
}


class SoftGroupHipError(RuntimeError):
    pass


def lib():
    """The loaded library.  Raises if it has not been built (python -m softgroup_amd.build)."""
    global _lib
    if _lib is None:
        # torch first: both link libamdhip64.so.7 and the process must end up with ONE HIP runtime
        # (the one torch ships), otherwise streams/pointers cross runtimes.
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise SoftGroupHipError(
                f'{LIB_PATH} not found: the HIP extension is required (no CPU fallback). '
                'Build it with `python -c "import __graft_entry__ as g; g.build()"`.')
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)   # AttributeError here = header/library mismatch: fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().sg_last_error()
        raise SoftGroupHipError(f'{what} failed ({rc}): {msg.decode() if msg else ""}')


def ptr(t):
    """Device/host pointer of a (contiguous) tensor, or NULL for None (argtypes turn the integer
    into a void*)."""
    return None if t is None else t.data_ptr()


_raw_stream = None


def stream():
    """handle of torch's current HIP stream on the current device (raw getter: this is called for
    every kernel launch, torch.cuda.current_stream() costs several microseconds)"""
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None) or (
            lambda dev: torch.cuda.current_stream(dev).cuda_stream)
    return _raw_stream(torch.cuda.current_device())


def workspace(nbytes, device):
    import torch
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
