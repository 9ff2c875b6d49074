"""ctypes binding of libgsraster.so -- the C-ABI declared in include/gsraster.h.

There is NO fallback: if the HIP library is missing or an entry point cannot be resolved, importing
this module raises.  Nothing here touches oracle/.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSRASTER_LIB") or os.path.join(_HERE, "libgsraster.so")  # env: experiment builds only

c_int, c_float, c_void_p, c_size_t, c_int64 = (ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t,
                                                ctypes.c_int64)

# name -> (restype, argtypes); must list every function of include/gsraster.h (tests/test_abi_cpu.py checks)
SIGNATURES = {
    "gsr_error_string": (ctypes.c_char_p, [c_int]),
    "gsr_abi_version": (c_int, []),
    "gsr_get_block_xy": (c_int, [ctypes.POINTER(c_int)] * 3),
    "gsr_preprocess_forward": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gsr_preprocess_backward": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float] +
                                [c_void_p] * 6 + [c_int] + [c_void_p] * 6),
    "gsr_get_local2j_ids_bool": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p]),
    "gsr_bin_prepare_bytes": (c_size_t, [c_int, c_int, c_int]),
    "gsr_bin_prepare": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_size_t, ctypes.POINTER(c_int64), c_void_p]),
    "gsr_bin_sort_bytes": (c_size_t, [c_int, c_int64, c_int, c_int]),
    "gsr_bin_sort": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_size_t, c_void_p,
                             c_void_p, c_void_p]),
    "gsr_bin_prepare_async": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_size_t, ctypes.POINTER(ctypes.c_uint32), c_void_p]),
    "gsr_set_depth_tie_order": (c_int, [c_int]),
    "gsr_bin_count_wait": (c_int, [ctypes.c_uint32, ctypes.POINTER(c_int64), c_void_p]),
    "gsr_bin_sort_capacity": (c_int64, [c_int, c_size_t, c_int, c_int]),
    "gsr_bin_sort_bounded": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_size_t, c_void_p,
                                     c_void_p, c_void_p]),
    "gsr_render_forward": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gsr_l1_ssim_num_partials": (c_int, [c_int, c_int, c_int]),
    "gsr_l1_ssim_forward": (c_int, [c_int, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p]),
    "gsr_l1_ssim_backward": (c_int, [c_int, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_float, c_float, c_void_p, c_int64, c_void_p]),
    "gsr_l1_ssim_forward_band": (c_int, [c_int, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p]),
    "gsr_l1_ssim_backward_band": (c_int, [c_int, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_float, c_float, c_void_p, c_int64, c_void_p, c_void_p]),
    "gsr_stamp": (c_int, [c_void_p, c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, c_void_p]),
    "gsr_band_mask": (c_int, [c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "gsr_l1_ssim_finalize": (c_int, [c_int, c_void_p, c_float, c_float, c_float, c_float, c_void_p, c_void_p]),
    "gsr_exchange_need": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p]),
    "gsr_exchange_chunks": (c_size_t, [c_int]),
    "gsr_exchange_count": (c_int, [c_int] * 7 + [c_void_p] * 6),
    "gsr_exchange_pack": (c_int, [c_int] * 9 + [c_void_p] * 8 + [c_int64, c_void_p, c_void_p, c_void_p]),
    "gsr_exchange_pack_slab": (c_int, [c_int] * 9 + [c_void_p] * 9 + [c_int64, c_void_p, c_void_p, c_void_p]),
    "gsr_exchange_unpack": (c_int, [c_int64] + [c_void_p] * 7),
    "gsr_zero_async": (c_int, [c_void_p, c_size_t, c_void_p]),
    "gsr_scatter_add_rows": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gsr_knn_workspace_bytes": (c_size_t, [c_int]),
    "gsr_knn_mean_dist2": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "gsr_densify_stats": (c_int, [c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gsr_group_rows_bytes": (c_size_t, [c_int64]),
    "gsr_group_rows": (c_int, [c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "gsr_gather_rows": (c_int, [c_int64, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p]),
    "gsr_scatter_rows": (c_int, [c_int64, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p]),
    "gsr_adam_step": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_double, ctypes.c_double,
                              ctypes.c_double, ctypes.c_double, c_int64, c_float, c_void_p]),
    "gsr_adam_step_multi": (c_int, [c_int] + [c_void_p] * 10 + [c_float, c_void_p]),
    "gsr_preprocess_forward_raw": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_float] + [c_void_p] * 7 +
                                   [c_int, c_int, c_float, c_float] + [c_void_p] * 8),
    "gsr_preprocess_backward_raw": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_float] + [c_void_p] * 7 +
                                    [c_int, c_int, c_float, c_float] + [c_void_p] * 6 + [c_int] + [c_void_p] * 7),
    "gsr_preprocess_forward_raw_batched": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float] +
                                           [c_void_p] * 5 + [c_int, c_int] + [c_void_p] * 8),
    "gsr_preprocess_backward_raw_batched": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float] +
                                            [c_void_p] * 5 + [c_int, c_int] + [c_void_p] * 6 + [c_int] +
                                            [c_void_p] * 7),
    "gsr_preprocess_backward_adam_raw_batched": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float] +
                                                 [c_void_p] * 5 + [c_int, c_int] + [c_void_p] * 6 + [c_int] +
                                                 [c_void_p] * 7 + [c_float, c_void_p, c_void_p]),
    "gsr_preprocess_backward_adam_raw_batched_dyn": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float] +
                                                     [c_void_p] * 5 + [c_int, c_int] + [c_void_p] * 6 + [c_int] +
                                                     [c_void_p] * 7 + [c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gsr_bin_total_offset": (c_size_t, [c_int, c_int, c_int]),
    "gsr_bin_segments_offset": (c_size_t, [c_int, c_int, c_int]),
    "gsr_bin_speculative": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_size_t, c_int64, c_void_p, c_size_t, c_void_p, c_void_p, ctypes.POINTER(c_int64),
                                    ctypes.POINTER(c_int), c_void_p]),
    "gsr_bin_speculative_async": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_size_t, c_int64, c_void_p, c_size_t, c_void_p, c_void_p,
                                          ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(c_int), c_void_p]),
    "gsr_set_bin_persistent": (c_int, [c_int]),
    "gsr_set_tile_cull": (c_int, [c_int]),
    "gsr_set_bin_rowmajor": (c_int, [c_int]),
    "gsr_bin_persist_status": (c_int, [ctypes.POINTER(ctypes.c_uint32)]),
    "gsr_bin_timeline": (c_int, [c_int, c_void_p, c_int, ctypes.POINTER(c_int)]),
    "gsr_flag_if_greater": (c_int, [c_void_p, ctypes.c_uint32, c_void_p, ctypes.c_uint32, c_void_p, c_void_p]),
    "gsr_render_seg_bytes": (c_size_t, [c_int, c_int]),
    "gsr_render_forward_seg": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int,
                                       c_void_p]),
    "gsr_render_backward_seg": (c_int, [c_int, c_int, c_int] + [c_void_p] * 13 + [c_size_t, c_int, c_int, c_void_p]),
    "gsr_render_forward_seg_z": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int,
                                         c_int, c_void_p, c_size_t, c_void_p]),
    "gsr_render_backward_seg_z": (c_int, [c_int, c_int, c_int] + [c_void_p] * 13 + [c_size_t, c_int, c_int, c_int,
                                                                                   c_void_p]),
    "gsr_composite_walked": (c_int, [ctypes.POINTER(ctypes.c_ulonglong), c_int]),
    "gsr_publish_flag": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_uint32, c_void_p]),
    "gsr_exchange_check": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, ctypes.c_uint64, c_int, c_void_p,
                                   ctypes.c_uint32, ctypes.c_uint32, c_void_p, c_void_p]),
    "gsr_activate_forward": (c_int, [c_int, c_int] + [c_void_p] * 10),
    "gsr_activate_backward": (c_int, [c_int, c_int] + [c_void_p] * 13),
    "gsr_render_backward": (c_int, [c_int, c_int, c_int] + [c_void_p] * 12),
}

ABI_VERSION = 13


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the gfx950 HIP library has not been built "
            "(run `python __graft_entry__.py build` or `python grendel-gs_amd/csrc/build.py`). "
            "There is no CPU or PyTorch fallback for this operator."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError -> loud failure on a stale library
        fn.restype = res
        fn.argtypes = args
    if lib.gsr_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {lib.gsr_abi_version()} != expected {ABI_VERSION}; rebuild")
    return lib


lib = _load()


def check(code, what):
    if code != 0:
        msg = lib.gsr_error_string(code)
        raise RuntimeError(f"{what} failed: {msg.decode() if msg else code} (code {code})")
