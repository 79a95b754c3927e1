"""ctypes binding of libcofi_hip.so (the C ABI declared in include/cofi_hip.h).

There is NO fallback: if the shared library is missing or does not load, importing the product
path raises.  `python -m cofii2p_amd.build` (or `__graft_entry__.build()`) produces the library
in-tree with hipcc for gfx950.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcofi_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "cofi_hip.h")
TUNE_HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "cofi_hip_tune.h")   # tuning / test hooks: tools/ and tests/ only

c_void_p, c_int, c_float, c_size_t = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
_P, _I, _F, _Z = c_void_p, c_int, c_float, c_size_t


class NormDesc(ctypes.Structure):
    """cofi_norm_desc_t (include/cofi_hip.h): a pending GroupNorm / InstanceNorm described by its statistics partials."""
    _fields_ = [("partials", c_void_p), ("nslab", c_int), ("width", c_int), ("channels", c_int), ("groups", c_int),
                ("gamma", c_void_p), ("beta", c_void_p), ("eps", c_float), ("slope", c_float), ("scale_shift", c_void_p), ("slab_rows", c_int)]


_N = ctypes.POINTER(NormDesc)


class TailDesc(ctypes.Structure):
    """cofi_loftr_tail_desc_t (include/cofi_hip.h): one LoFTR layer tail with its optional fused successors."""
    _fields_ = [("msg", c_void_p), ("ldm", c_int), ("rows", c_int),
                ("parts", c_void_p), ("parts_bytes", c_size_t), ("L", c_int), ("S", c_int), ("H", c_int), ("frames", c_int),
                ("x", c_void_p), ("ldx", c_int), ("planes", c_int),
                ("wm", c_void_p), ("w0", c_void_p), ("w2", c_void_p),
                ("n1_gamma", c_void_p), ("n1_beta", c_void_p), ("n2_gamma", c_void_p), ("n2_beta", c_void_p), ("eps", c_float),
                ("out", c_void_p), ("ldo", c_int),
                ("proj_n", c_int * 2), ("proj_w", c_void_p * 2), ("proj_y", c_void_p * 2), ("proj_ldy", c_int * 2), ("proj_part", c_void_p * 2),
                ("out_l2", c_void_p), ("ld_l2", c_int), ("out_l2t", c_void_p), ("ld_l2t", c_int), ("w_frag", c_int)]


_T = ctypes.POINTER(TailDesc)

ABI_VERSION = 3   # COFI_ABI_VERSION of include/cofi_hip.h this table mirrors

# name -> (restype, argtypes); mirrors include/cofi_hip.h declaration by declaration
SIGNATURES = {
    "cofi_abi_version": (_I, []),
    "cofi_target_arch": (ctypes.c_char_p, []),
    "cofi_knn_topk": (_I, [_P, _I, _P, _I, _I, _P, _P, _P]),
    "cofi_knn_up_nearest": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _I, _P]),
    "cofi_knn_grid_workspace": (_Z, [_I]),
    "cofi_knn_grid_build": (_I, [_P, _I, _P, _Z, _P, _P]),
    "cofi_knn_topk_grid": (_I, [_P, _Z, _I, _P, _P, _I, _I, _P, _P, _P]),
    "cofi_nearest_node": (_I, [_P, _I, _P, _I, _P, _P]),
    "cofi_nearest_node_sel": (_I, [_P, _I, _P, _P, _P, _I, _P, _P]),
    "cofi_idx64_to_idx32": (_I, [_P, _P, _Z, _P]),
    "cofi_idx32_to_idx64": (_I, [_P, _P, _Z, _P]),
    "cofi_row_sum_positive": (_I, [_P, _I, _I, _I, _P, _P]),
    "cofi_kp_pack_c4": (_I, [_P, _I, _I, _P, _I, _P, _P]),
    "cofi_kpconv_aggregate_c4": (_I, [_P, _I, _I, _P, _P, _I, _I, _P, _F, _P, _I, _P, _I, _P, _P]),
    "cofi_kpconv_aggregate": (_I, [_P, _I, _I, _I, _P, _P, _P, _I, _I, _P, _F, _P, _P, _I, _I, _P, _I, _P, _P]),
    "cofi_neighbor_maxpool": (_I, [_P, _I, _I, _I, _P, _I, _I, _P, _I, _I, _P, _P]),
    "cofi_gather_rows": (_I, [_P, _I, _I, _I, _P, _I, _I, _P, _I, _I, _P]),
    "cofi_gemm_f16x3_eligible": (_I, [_I, _I, _I, _I, _I]),
    "cofi_conv2d_f16x3_eligible": (_I, [_I, _I, _I, _I, _I, _I, _I, _I, _I, _I]),
    "cofi_gemm_f32_workspace": (_Z, [_I, _I, _I]),
    "cofi_gemm_f32": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _I, _P, _Z, _P]),
    "cofi_split_bf16_planes": (_I, [_P, _I, _I, _I, _P, _I, _I, _P]),
    "cofi_gemm_f32_stat_slabs": (_I, [_I, _I, _I]),
    "cofi_gemm_f32_colstats": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _I, _P, _P, _Z, _P]),
    "cofi_gemm_f32_fused": (_I, [_P, _I, _N, _P, _I, _P, _I, _I, _I, _I, _P, _P, _I, _P, _I, _P, _Z, _I, _P]),
    "cofi_gemm_f32_layernorm": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _F, _I, _P, _I, _P, _Z, _P]),
    "cofi_group_stats_from_colpart": (_I, [_P, _I, _I, _I, _I, _F, _P, _I, _P]),
    "cofi_col_inv_norm_from_colpart": (_I, [_P, _I, _I, _I, _I, _F, _P, _I, _P]),
    "cofi_group_stats_workspace": (_Z, [_I, _I, _I, _I]),
    "cofi_group_stats": (_I, [_P, _I, _I, _I, _I, _F, _P, _P, _Z, _I, _P]),
    "cofi_group_stats_exact": (_I, [_P, _I, _I, _I, _I, _F, _P, _P, _Z, _I, _P]),
    "cofi_group_norm_apply": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _F, _P, _I, _P, _I, _P]),
    "cofi_norm_finalize": (_I, [_N, _I, _I, _P, _P]),
    "cofi_group_norm_apply_partials": (_I, [_P, _I, _I, _I, _N, _P, _I, _N, _P, _I, _P, _I, _P]),
    "cofi_layer_norm": (_I, [_P, _I, _I, _I, _P, _P, _F, _I, _P, _I, _P, _I, _P]),
    "cofi_layer_norm_act": (_I, [_P, _I, _I, _I, _P, _P, _F, _F, _P, _I, _I, _P, _I, _P]),
    "cofi_loftr_tail_bf16x3": (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _I, _I, _P]),
    "cofi_loftr_tail": (_I, [_T, _P]),
    "cofi_loftr_tail_parts_bf16x3": (_I, [_P, _Z, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _I, _P]),
    "cofi_attention_workspace": (_Z, [_I, _I, _I, _I, _I]),
    "cofi_attention_parts": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _F, _I, _I, _I, _I, _F, _I, _P, _Z, _P]),
    "cofi_attention_parts_bf16x6": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _F, _I, _I, _I, _I, _F, _I, _P, _Z, _P]),
    "cofi_attention_kv_planes_bytes": (_Z, [_I, _I, _I, _I]),
    "cofi_attention_kv_planes": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "cofi_attention_parts_planes": (_I, [_P, _I, _P, _Z, _P, _P, _I, _I, _F, _I, _I, _I, _I, _F, _I, _P, _Z, _P]),
    "cofi_attention_merge": (_I, [_P, _Z, _I, _I, _I, _I, _I, _P, _I, _P]),
    "cofi_attention_fwd": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _F, _P, _Z, _I, _P]),
    "cofi_attention_fwd_colpart": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _F, _P, _I, _I, _I, _I, _I, _F, _P, _Z, _I, _P]),
    "cofi_col_inv_norm": (_I, [_P, _I, _I, _I, _F, _P, _P]),
    "cofi_pos_sine": (_I, [_P, _I, _I, _I, _P, _I, _I, _I, _P, _I, _P]),
    "cofi_l2norm_rows": (_I, [_P, _I, _I, _I, _P, _I, _I, _P]),
    "cofi_l2norm_rows2": (_I, [_P, _I, _I, _I, _P, _I, _P, _I, _P]),
    "cofi_transpose": (_I, [_P, _I, _I, _I, _P, _I, _I, _P]),
    "cofi_col_mean": (_I, [_P, _I, _I, _I, _P, _I, _P]),
    "cofi_conv2d_nhwc": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _P, _P, _I, _I, _P, _I, _P, _P, _Z, _I, _P]),
    "cofi_conv2d_nhwc_fused": (_I, [_P, _I, _N, _I, _I, _I, _P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _P, _I, _P, _I, _P, _Z, _I, _P]),
    "cofi_im2col_stem": (_I, [_P, _I, _I, _I, _P, _I, _P]),
    "cofi_maxpool3x3s2_nhwc": (_I, [_P, _I, _I, _I, _P, _I, _P]),
    "cofi_upsample2x_cat_nhwc": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _P, _I, _I, _P]),
    "cofi_extract_patches_nhwc": (_I, [_P, _I, _I, _I, _I, _P, _I, _F, _P, _I, _P, _P]),
    "cofi_row_argmin_1m": (_I, [_P, _I, _I, _I, _P, _P]),
    "cofi_select_matches": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _I, _I, _P, _P, _P, _I, _P]),
    "cofi_gather_points_sel": (_I, [_P, _P, _P, _I, _P, _P]),
    "cofi_gather_rows_sel": (_I, [_P, _I, _I, _P, _P, _I, _P, _I, _P]),
    "cofi_multi_copy": (_I, [_P, _I, _I, _P]),
    "cofi_pnp_ransac_workspace": (_Z, [_I]),
    "cofi_desc_loss_workspace": (_Z, [_I]),
    "cofi_desc_loss": (_I, [_P, _I, _P, _I, _P, _I, _I, _F, _F, _F, _P, _P, _P, _P, _I, _P, _I, _P, _Z, _P]),
    "cofi_fine_circle_loss": (_I, [_P, _P, _I, _P, _I, _I, _F, _F, _P, _P, _P, _P, _P, _I, _P]),
    "cofi_overlap_loss": (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _P]),
    "cofi_kpconv_aggregate_bwd": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _P, _F, _P, _I, _P]),
    "cofi_neighbor_maxpool_arg": (_I, [_P, _I, _I, _I, _P, _I, _I, _P, _I, _P, _P]),
    "cofi_neighbor_maxpool_bwd": (_I, [_P, _I, _P, _I, _I, _P, _P, _I, _P, _I, _P]),
    "cofi_gather_rows_bwd": (_I, [_P, _I, _I, _P, _P, _I, _P, _I, _P]),
    "cofi_im2col_nhwc": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P]),
    "cofi_col2im_nhwc": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P]),
    "cofi_group_norm_bwd_workspace": (_Z, [_I, _I, _I]),
    "cofi_group_norm_bwd": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _F, _I, _P, _I, _P, _P, _P, _I, _P, _Z, _P]),
    "cofi_col_sum_workspace": (_Z, [_I, _I]),
    "cofi_col_sum": (_I, [_P, _I, _I, _I, _P, _P, _Z, _P]),
    "cofi_l2norm_rows_bwd": (_I, [_P, _I, _P, _I, _I, _I, _F, _P, _I, _P]),
    "cofi_upsample2x_bwd_nhwc": (_I, [_P, _I, _I, _I, _I, _P, _I, _P]),
    "cofi_transpose_pair": (_I, [_P, _I, _I, _P, _I, _P, _I, _I, _P, _I, _I, _P]),
    "cofi_col_normalize_workspace": (_Z, [_I, _I]),
    "cofi_col_normalize": (_I, [_P, _I, _P, _I, _I, _I, _F, _I, _P, _P, _I, _P, _Z, _P]),
    "cofi_attention_bwd_workspace": (_Z, [_I, _I]),
    "cofi_attention_bwd": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _F, _P, _I, _P, _I, _P, _I, _P, _Z, _P]),
    "cofi_pnp_ransac": (_I, [_P, _P, _P, _I, _F, _F, _F, _F, _I, _F, ctypes.c_uint, _I, _P, _Z, _P, _P, _P, _P]),
    "cofi_match_finish": (_I, [_P, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _P, _I, _F, _P, _I, _P, _P, _P, _I, _P, _P, _I, _I, _P]),
    "cofi_fine_match": (_I, [_P, _P, _I, _I, _P, _I, _F, _P, _I, _P, _P, _P]),
    "cofi_kpconv_fused_slab_rows": (_I, [_I, _I, _I]),
    "cofi_kpconv_fused": (_I, [_P, _I, _I, _I, _P, _P, _P, _I, _I, _P, _F, _P, _P, _I, _P, _P, _I, _P, _I, _I, _P, _P]),
    "cofi_color_jitter_chw": (_I, [_P, _I, _I, _P, _F, _F, _F, _F, _P, _Z, _P]),
    "cofi_pack_transform_scan": (_I, [_P, _I, _P, _P, _P]),
    "cofi_voxel_downsample_workspace": (_Z, [_I]),
    "cofi_voxel_downsample": (_I, [_P, _I, ctypes.c_double, _P, _I, _P, _P, _Z, _P]),
    "cofi_grid_subsample": (_I, [_P, _I, _F, _P, _I, _P, _P, _Z, _P]),
    "cofi_radius_mask": (_I, [_P, _P, _I, _I, _I, _F, ctypes.c_longlong, ctypes.c_longlong, _P, _P, _P]),
    "cofi_gather_transform": (_I, [_P, _P, _I, _P, _P, _P, _I, _P]),
    "cofi_resize_crop_image": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
}


# include/cofi_hip_tune.h: plan overrides of the calling thread for tools/ and tests/ - not bound by load(), not used by the product
TUNE_SIGNATURES = {
    "cofi_tune_force_plan": (_I, [_I, _I, _I]),
    "cofi_tune_force_planes": (_I, [_I, _I]),
    "cofi_tune_force_big": (_I, [_I, _I]),
    "cofi_tune_force_conv_direct": (_I, [_I]),
    "cofi_tune_big_debug": (_I, [_I]),
    "cofi_tune_f16x3_resplit_events": (ctypes.c_long, [_I]),
    "cofi_tune_f16x3_launch_flops": (ctypes.c_long, [_I, _P]),
}


def header_symbols(path=None):
    """Every function name declared in include/cofi_hip.h (or in the header at `path`)."""
    text = open(path or HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cofi_[a-z0-9_]+)\s*\(", text)))


class CofiError(RuntimeError):
    pass


_lib = None


def load():
    """Returns the loaded library; raises CofiError if it is missing (never falls back)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CofiError("libcofi_hip.so not found at %s — build it with `python -m cofii2p_amd.build` "
                        "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise CofiError("libcofi_hip.so failed to load: %s" % e)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise CofiError("libcofi_hip.so does not export %s (stale build?)" % name)
        fn.restype = res
        fn.argtypes = args
    if lib.cofi_abi_version() != ABI_VERSION:
        raise CofiError("ABI version mismatch: libcofi_hip.so reports %d, this binding mirrors include/cofi_hip.h version %d" % (lib.cofi_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        kind = {-1: "COFI_EINVAL", -2: "COFI_EWORKSPACE", -3: "COFI_EUNSUPPORTED"}.get(rc, "hipError %d" % rc)
        raise CofiError("%s failed: %s" % (what, kind))
