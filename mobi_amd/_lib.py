"""ctypes binding of libmobi_hip.so (include/mobi_engine.h).

The product path has NO fallback: if the library cannot be loaded every engine
call raises `EngineUnavailable` ("the product path must fail
loudly when the HIP extension is missing").
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmobi_hip.so")

MOBI_F16, MOBI_BF16 = 0, 1
ABI_VERSION = 6            # include/mobi_engine.h MOBI_ABI_VERSION
EPI_NONE, EPI_GEGLU = 0, 1
OUT_ROWS, OUT_TRANSPOSED, OUT_ROWS_F32 = 0, 1, 2
ACT_NONE, ACT_SILU, ACT_GELU = 0, 1, 2

vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class EngineUnavailable(RuntimeError):
    pass


class EngineError(RuntimeError):
    pass


class IgemmParams(C.Structure):
    _fields_ = [("src0", vp), ("src1", vp), ("c0", i32), ("c1", i32), ("batch", i32), ("hin", i32), ("win", i32),
                ("upsample", i32), ("hout", i32), ("wout", i32), ("kh", i32), ("kw", i32), ("stride", i32),
                ("pad_h", i32), ("pad_w", i32), ("src_img_stride", i64), ("weight", vp), ("groups", i32),
                ("w_group_stride", i64), ("n_packed", i32), ("cout", i32), ("bias", vp), ("rowvec", vp), ("rowvec_stride", i32),
                ("residual", vp), ("res_img_stride", i64), ("out", vp), ("out_img_stride", i64),
                ("out_mode", i32), ("epilogue", i32), ("scale", f32), ("dtype", i32), ("split_k", i32), ("ws", vp), ("k_order", i32),
                ("weight_tiled", vp), ("sync", vp), ("ln_svec", vp), ("ln_eps", f32), ("defer_finish", i32)]


class SplitSource(C.Structure):
    _fields_ = [("slabs", vp), ("count", i32), ("row_stride", i32), ("bias", vp), ("rowvec", vp), ("rowvec_stride", i32),
                ("residual", vp), ("res_img_stride", i64), ("finished", vp)]


class GroupNormParams(C.Structure):
    _fields_ = [("src0", vp), ("src1", vp), ("c0", i32), ("c1", i32), ("batch", i32), ("hw", i32), ("gamma", vp),
                ("beta", vp), ("eps", f32), ("silu", i32), ("out", vp), ("ws", vp), ("dtype", i32), ("src_f32", i32),
                ("out_mode", i32), ("sync", vp), ("src0_split", C.POINTER(SplitSource))]


class LayerNormParams(C.Structure):
    _fields_ = [("src", vp), ("out", vp), ("images", i32), ("rows_per_image", i32), ("channels", i32),
                ("src_img_stride", i64), ("out_img_stride", i64), ("gamma", vp), ("beta", vp), ("eps", f32),
                ("dtype", i32)]


class AttentionParams(C.Structure):
    _fields_ = [("q", vp), ("q_img_stride", i64), ("q_row_stride", i32),
                ("k", vp), ("k_img_stride", i64), ("k_row_stride", i32),
                ("vt", vp), ("vt_img_stride", i64), ("vt_row_stride", i32),
                ("out", vp), ("out_img_stride", i64), ("out_row_stride", i32),
                ("images", i32), ("heads", i32), ("dh", i32), ("tq", i32), ("tk", i32), ("scale", f32),
                ("dtype", i32), ("v_layout", i32), ("q_log2_scaled", i32)]


class FfGegluParams(C.Structure):
    _fields_ = [("x", vp), ("rows", i64), ("c", i32), ("hidden", i32), ("w_packed", vp), ("b2", vp), ("residual", vp),
                ("out", vp), ("dtype", i32), ("ln_gamma", vp), ("ln_beta", vp), ("ln_eps", f32)]


CH_LOAD_S, CH_LOAD_R, CH_AFFINE_S, CH_ROWSTATS, CH_PRODUCT, CH_ADAPTER, CH_STORE_S = range(7)
CH_FOLD, CH_RESID, CH_TO_S, CH_STORE = 1, 2, 4, 8
CHAIN_MAX_OPS = 10


class ChainOp(C.Structure):
    _fields_ = [("code", i32), ("flags", i32), ("p0", vp), ("p1", vp), ("bias", vp), ("svec", vp), ("dst", vp),
                ("img_stride", i64), ("row_stride", i64), ("dst_img_stride", i64), ("dst_row_stride", i64),
                ("bias_img_stride", i64), ("img_div", i32), ("dst_img_div", i32), ("bias_img_div", i32), ("eps", f32)]


class RowChainParams(C.Structure):
    _fields_ = [("dtype", i32), ("channels", i32), ("images", i32), ("rows_per_image", i32), ("nprog", i32),
                ("nops", i32 * 2), ("prog", (ChainOp * CHAIN_MAX_OPS) * 2), ("ad_image", vp), ("ad_eps", f32)]


class LayerNormBwdParams(C.Structure):
    _fields_ = [("x", vp), ("dy", vp), ("x_row_stride", i64), ("dy_row_stride", i64), ("gamma", vp), ("eps", f32),
                ("dx_add", vp), ("dx", vp), ("partial", vp), ("dgamma_dbeta", vp), ("rows", i64), ("channels", i32),
                ("dtype", i32)]


class AttentionBwdParams(C.Structure):
    _fields_ = [("q", vp), ("q_img_stride", i64), ("q_row_stride", i64), ("k", vp), ("k_img_stride", i64), ("k_row_stride", i64),
                ("v", vp), ("v_img_stride", i64), ("v_row_stride", i64), ("o", vp), ("o_img_stride", i64), ("o_row_stride", i64),
                ("dout", vp), ("dout_img_stride", i64), ("dout_row_stride", i64), ("dq", vp), ("dk", vp), ("dv", vp),
                ("lse", vp), ("dvec", vp), ("images", i32), ("heads", i32), ("dh", i32), ("tq", i32), ("tk", i32),
                ("scale", f32), ("dtype", i32), ("force_vector", i32)]


class CtxAttentionParams(C.Structure):
    _fields_ = [("q", vp), ("out", vp), ("k", vp), ("v", vp), ("images", i32), ("heads", i32), ("dh", i32),
                ("tq", i32), ("tk", i32), ("scale", f32), ("dtype", i32)]


class TwoKeyAdapterParams(C.Structure):
    _fields_ = [("x", vp), ("out", vp), ("x_img_stride", i64), ("out_img_stride", i64), ("a", vp), ("a_sum", vp),
                ("c", vp), ("u", vp), ("b", vp), ("images", i32), ("rows_per_image", i32), ("channels", i32),
                ("heads", i32), ("eps", f32), ("dtype", i32), ("ln_out", vp * 2), ("ln_gamma", vp * 2), ("ln_beta", vp * 2),
                ("ln_eps", f32)]


class SkinnyLinearParams(C.Structure):
    _fields_ = [("x", vp), ("m", i32), ("k", i32), ("x_row_stride", i32), ("weight", vp), ("bias", vp),
                ("out", vp), ("n", i32), ("out_row_stride", i32), ("pre_act", i32), ("post_act", i32),
                ("dtype", i32)]


class ConvSmallCinParams(C.Structure):
    _fields_ = [("src", vp * 3), ("c", i32 * 3), ("batch", i32), ("h", i32), ("w", i32), ("kh", i32), ("kw", i32),
                ("pad_h", i32), ("pad_w", i32), ("weight", vp), ("bias", vp), ("cout", i32), ("out", vp),
                ("out_f32_nchw", i32), ("dtype", i32)]


class ConvSmallCoutParams(C.Structure):
    _fields_ = [("src", vp), ("cin", i32), ("batch", i32), ("h", i32), ("w", i32), ("kh", i32), ("kw", i32),
                ("pad_h", i32), ("pad_w", i32), ("weight", vp), ("bias", vp), ("cout", i32), ("out", vp),
                ("clamp", i32), ("clamp_lo", f32), ("clamp_hi", f32), ("in_scale", f32), ("dtype", i32)]


class DdimStepParams(C.Structure):
    _fields_ = [("x", vp), ("e_cond", vp), ("e_uncond", vp), ("noise", vp), ("x_prev", vp), ("pred_x0", vp),
                ("e_out", vp), ("n", i64), ("cfg_scale", f32), ("a_t", f32), ("a_prev", f32), ("sigma_t", f32),
                ("sqrt_one_minus_at", f32), ("temperature", f32), ("coef_dev", vp)]


class RangePasteParams(C.Structure):
    _fields_ = [("sample_depth", vp), ("sample_int", vp), ("depth_orig", vp), ("int_orig", vp), ("pitch", vp), ("yaw", vp),
                ("gt_mask", vp), ("planes", vp), ("crop_left", vp), ("width_crop", vp), ("depth_unc", vp), ("int_unc", vp),
                ("depth_final", vp), ("int_final", vp), ("pred_mask", vp), ("batch", i32), ("hc", i32), ("wc", i32),
                ("h0", i32), ("w0", i32), ("depth_min", f32), ("depth_max", f32)]


class LidarMetricsParams(C.Structure):
    _fields_ = [("pred", vp), ("gt", vp), ("inst_mask", vp), ("box_mask", vp), ("width_crop", vp), ("out", vp),
                ("batch", i32), ("h", i32), ("w", i32), ("pool_h", i32), ("max_width", i32)]


class RangePrepareParams(C.Structure):
    _fields_ = [("depth_orig", vp), ("int_orig", vp), ("inst_orig", vp), ("crop_left", vp), ("width_crop", vp),
                ("min_depth", vp), ("max_depth", vp), ("edit_mask", vp), ("range_data", vp), ("range_data_inpaint", vp),
                ("inst_out", vp), ("batch", i32), ("h0", i32), ("w0", i32), ("height", i32), ("width", i32),
                ("alpha", f32), ("object_norm", i32), ("int_norm", i32)]


class ImagePrepareParams(C.Structure):
    _fields_ = [("frames", vp), ("corners_xy", vp), ("invert", vp), ("crop", vp), ("gt", vp), ("inpaint", vp), ("mask", vp),
                ("batch", i32), ("H", i32), ("W", i32), ("height", i32), ("width", i32)]


STRUCT_IDS = {0: IgemmParams, 1: GroupNormParams, 2: LayerNormParams, 3: AttentionParams, 4: CtxAttentionParams,
              5: SkinnyLinearParams, 6: ConvSmallCinParams, 7: ConvSmallCoutParams, 8: DdimStepParams, 9: TwoKeyAdapterParams,
              10: RangePasteParams, 11: LidarMetricsParams, 12: RangePrepareParams, 13: ImagePrepareParams,
              14: FfGegluParams, 15: RowChainParams, 16: ChainOp,
              17: LayerNormBwdParams, 18: AttentionBwdParams, 19: SplitSource}

# every symbol include/mobi_engine.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "mobi_abi_version": (C.c_int, []),
    "mobi_error_string": (C.c_char_p, [C.c_int]),
    "mobi_struct_size": (C.c_size_t, [C.c_int]),
    "mobi_tuning_reload": (C.c_int, []),
    "mobi_build_info": (C.c_int, []),
    "mobi_igemm": (C.c_int, [C.POINTER(IgemmParams), vp]),
    "mobi_igemm_plan_splits": (C.c_int, [C.POINTER(IgemmParams)]),
    "mobi_igemm_kernel_variant": (C.c_int, [C.POINTER(IgemmParams)]),
    "mobi_igemm_workspace_bytes": (C.c_size_t, [C.POINTER(IgemmParams), i32]),
    "mobi_igemm_sync_bytes": (C.c_size_t, [C.POINTER(IgemmParams), i32]),
    "mobi_igemm_slab_count": (i32, [C.POINTER(IgemmParams)]),
    "mobi_igemm_finish": (C.c_int, [C.POINTER(IgemmParams), vp]),
    "mobi_groupnorm_takes_split": (C.c_int, [i32, i32, i32, i32]),
    "mobi_groupnorm_workspace_bytes": (C.c_size_t, [i32, i32]),
    "mobi_groupnorm": (C.c_int, [C.POINTER(GroupNormParams), vp]),
    "mobi_layernorm": (C.c_int, [C.POINTER(LayerNormParams), vp]),
    "mobi_attention": (C.c_int, [C.POINTER(AttentionParams), vp]),
    "mobi_ctx_attention": (C.c_int, [C.POINTER(CtxAttentionParams), vp]),
    "mobi_two_key_adapter": (C.c_int, [C.POINTER(TwoKeyAdapterParams), vp]),
    "mobi_two_key_adapter_fuses_ln": (C.c_int, [i32, i64]),
    "mobi_softmax_rows": (C.c_int, [vp, vp, i64, i32, i32, vp]),
    "mobi_skinny_linear": (C.c_int, [C.POINTER(SkinnyLinearParams), vp]),
    "mobi_layernorm_rows_f32": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, f32, vp]),
    "mobi_linear_f32": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "mobi_ff_geglu": (C.c_int, [C.POINTER(FfGegluParams), vp]),
    "mobi_ff_geglu_packed_bytes": (C.c_size_t, [i32, i32]),
    "mobi_row_chain": (C.c_int, [C.POINTER(RowChainParams), vp]),
    "mobi_groupnorm_scale_shift": (C.c_int, [vp, vp, vp, f32, vp, vp, i32, i32, i32, i32, vp]),
    "mobi_row_chain_weight_bytes": (C.c_size_t, [i32]),
    "mobi_row_chain_supported": (C.c_int, [i32, i32]),
    "mobi_row_chain_adapter_image_bytes": (C.c_size_t, [i32]),
    "mobi_row_chain_adapter_image": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "mobi_trunk_add": (C.c_int, [vp, vp, vp, i64, i32, vp]),
    "mobi_split_f32": (C.c_int, [vp, vp, i64, i32, i32, i32, vp]),
    "mobi_transpose": (C.c_int, [vp, i64, vp, i32, i32, i32, vp]),
    "mobi_tile_weights": (C.c_int, [vp, vp, i32, i32, vp]),
    "mobi_backward_partial_blocks": (i32, [i64]),
    "mobi_colsum": (C.c_int, [vp, i64, i64, i32, i32, vp, vp, vp]),
    "mobi_layernorm_bwd": (C.c_int, [C.POINTER(LayerNormBwdParams), vp]),
    "mobi_geglu_fwd": (C.c_int, [vp, vp, i64, i32, i32, vp]),
    "mobi_geglu_bwd": (C.c_int, [vp, vp, vp, i64, i32, i32, vp]),
    "mobi_attention_bwd": (C.c_int, [C.POINTER(AttentionBwdParams), vp]),
    "mobi_groupnorm_bwd_workspace_floats": (C.c_size_t, [i32, i32, i32]),
    "mobi_groupnorm_bwd": (C.c_int, [vp, vp, vp, vp, f32, i32, vp, vp, i32, i32, i32, i32, vp, vp]),
    "mobi_sumpool2": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "mobi_add": (C.c_int, [vp, vp, vp, i64, i32, vp]),
    "mobi_silu_bwd_f32": (C.c_int, [vp, vp, vp, i64, vp]),
    "mobi_adamw_step": (C.c_int, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, vp]),
    "mobi_quick_gelu": (C.c_int, [vp, vp, i64, i32, vp]),
    "mobi_timestep_embedding": (C.c_int, [vp, vp, vp, i32, i32, vp]),
    "mobi_conv_small_cin": (C.c_int, [C.POINTER(ConvSmallCinParams), vp]),
    "mobi_conv_small_cout": (C.c_int, [C.POINTER(ConvSmallCoutParams), vp]),
    "mobi_ddim_step": (C.c_int, [C.POINTER(DdimStepParams), vp]),
    "mobi_lincomb4": (C.c_int, [vp, vp, vp, vp, vp, f32, f32, f32, f32, i64, vp]),
    "mobi_q_sample": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "mobi_mask_blend": (C.c_int, [vp, vp, vp, vp, f32, f32, i32, i32, i32, vp]),
    "mobi_posterior_sample": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, i32, f32, vp]),
    "mobi_range_denorm": (C.c_int, [vp, vp, vp, f32, f32, f32, f32, i32, i32, vp, vp, i32, i32, vp]),
    "mobi_range_paste": (C.c_int, [C.POINTER(RangePasteParams), vp]),
    "mobi_lidar_metrics": (C.c_int, [C.POINTER(LidarMetricsParams), vp]),
    "mobi_range_prepare": (C.c_int, [C.POINTER(RangePrepareParams), vp]),
    "mobi_box_mask": (C.c_int, [vp, vp, vp, i32, i32, i32, vp]),
    "mobi_image_prepare": (C.c_int, [C.POINTER(ImagePrepareParams), vp]),
    "mobi_paste_patch": (C.c_int, [vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, vp]),
    "mobi_gaussian_blur": (C.c_int, [vp, vp, vp, i32, i32, vp, i32, vp]),
    "mobi_blend_frame": (C.c_int, [vp, vp, vp, vp, i32, i32, vp]),
    "mobi_nearest_resize": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "mobi_pack_nchw_sources": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp]),
    "mobi_nchw_f32_to_nhwc": (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
    "mobi_nhwc_to_nchw_f32": (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
}

_lib = None


def load():
    """Load the library once, bind every symbol, verify struct layouts."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineUnavailable(f"{LIB_PATH} not built: run `python -m mobi_amd.build` (needs hipcc). "
                                "The MObI engine has no CPU or PyTorch fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise EngineUnavailable(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise EngineUnavailable(f"{LIB_PATH} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    if lib.mobi_abi_version() != ABI_VERSION:
        raise EngineUnavailable(f"ABI mismatch: {LIB_PATH} is version {lib.mobi_abi_version()}, the binding {ABI_VERSION} "
                                "(include/mobi_engine.h MOBI_ABI_VERSION): rebuild with `python -m mobi_amd.build --force`")
    for sid, cls in STRUCT_IDS.items():
        want = lib.mobi_struct_size(sid)
        if want != C.sizeof(cls):
            raise EngineUnavailable(f"ABI mismatch: {cls.__name__} is {C.sizeof(cls)} bytes in the binding, "
                                    f"{want} in the library")
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        raise EngineError(f"{what} failed: {load().mobi_error_string(code).decode()} ({code})")
