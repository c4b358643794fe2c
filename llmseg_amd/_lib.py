"""ctypes binding of libllmseg_hip.so (C ABI declared in include/llmseg_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a symbol is
absent, importing/using the ops raises.  Build with `llmseg_amd/csrc/build.sh`
(`__graft_entry__.build()` does that).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LLMSEG_LIB") or os.path.join(_HERE, "libllmseg_hip.so")     # LLMSEG_LIB: side builds of the same ABI (tools/ experiments)

ABI_VERSION = 9          # == LLMSEG_ABI_VERSION of include/llmseg_hip.h

ACT_NONE, ACT_RELU, ACT_GELU, ACT_QUICKGELU, ACT_SILU, ACT_SIGMOID = range(6)


class _Sized(C.Structure):
    """Argument struct that starts with the ABI guard: `struct_size` = sizeof of THIS declaration, filled in on construction."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_size = C.sizeof(self)


class GemmArgs(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("reserved0", C.c_uint32), ("A", C.c_void_p), ("W", C.c_void_p), ("C", C.c_void_p),
                ("bias", C.c_void_p), ("gamma", C.c_void_p), ("residual", C.c_void_p),
                ("M", C.c_int64), ("N", C.c_int64), ("K", C.c_int64),
                ("lda", C.c_int64), ("ldw", C.c_int64), ("ldc", C.c_int64), ("ldr", C.c_int64),
                ("batch", C.c_int64), ("strideA", C.c_int64), ("strideW", C.c_int64), ("strideC", C.c_int64),
                ("alpha", C.c_float), ("act", C.c_int), ("out_f32", C.c_int), ("trans_a", C.c_int), ("trans_w", C.c_int),
                ("batch2", C.c_int64), ("strideA2", C.c_int64), ("strideW2", C.c_int64), ("strideC2", C.c_int64),
                ("A2", C.c_void_p), ("W2", C.c_void_p), ("lda2", C.c_int64), ("ldw2", C.c_int64),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64), ("accumulate", C.c_int),
                ("a_norm_w", C.c_void_p), ("a_norm_eps", C.c_float), ("a_swiglu", C.c_int),
                ("norm_w", C.c_void_p), ("norm_eps", C.c_float), ("reserved1", C.c_int), ("norm_out", C.c_void_p), ("ldn", C.c_int64),
                ("fx", C.c_int), ("fx_T", C.c_int), ("fx_cols", C.c_int64), ("fx_cos", C.c_void_p), ("fx_sin", C.c_void_p), ("fx_out", C.c_void_p),
                ("fx_in", C.c_void_p), ("fx_ld", C.c_int64),
                ("nb_x", C.c_void_p), ("nb_w", C.c_void_p), ("nb_dres", C.c_void_p), ("nb_eps", C.c_float), ("nb_rms", C.c_int),
                ("nb_lora_t", C.c_void_p), ("nb_lora_ldt", C.c_int64), ("nb_lora_w0", C.c_void_p), ("nb_lora_w1", C.c_void_p), ("nb_lora_alpha", C.c_float),
                ("reserved2", C.c_int), ("nb_lora_drop", C.c_void_p),
                ("dl_o", C.c_void_p), ("dl_ldo", C.c_int64), ("dl_out", C.c_void_p), ("dl_heads", C.c_int32), ("dl_T", C.c_int32),
                ("nb_lora_part", C.c_void_p), ("nb_lora_S", C.c_int32), ("nb_lora_scale", C.c_float), ("nb_lora_zero", C.c_int32), ("reserved3", C.c_int32)]

FX_NONE, FX_ROPE, FX_SWIGLU, FX_SWIGLU_BWD = 0, 1, 2, 3          # LLMSEG_FX_* of include/llmseg_hip.h


class AttnArgs(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("reserved0", C.c_uint32), ("Q", C.c_void_p), ("K", C.c_void_p), ("V", C.c_void_p), ("O", C.c_void_p),
                ("q_stride_b", C.c_int64), ("q_stride_h", C.c_int64), ("q_stride_row", C.c_int64),
                ("k_stride_b", C.c_int64), ("k_stride_h", C.c_int64), ("k_stride_row", C.c_int64),
                ("v_stride_b", C.c_int64), ("v_stride_h", C.c_int64), ("v_stride_row", C.c_int64),
                ("o_stride_b", C.c_int64), ("o_stride_h", C.c_int64), ("o_stride_row", C.c_int64),
                ("batch", C.c_int32), ("heads", C.c_int32), ("Nq", C.c_int32), ("Nk", C.c_int32), ("head_dim", C.c_int32),
                ("scale", C.c_float), ("causal", C.c_int32),
                ("key_mask", C.c_void_p), ("rel_h", C.c_void_p), ("rel_w", C.c_void_p),
                ("rel_ld", C.c_int32), ("grid_h", C.c_int32), ("grid_w", C.c_int32),
                ("o_row_map", C.c_void_p), ("rel_tab_h", C.c_void_p), ("rel_tab_w", C.c_void_p), ("lse", C.c_void_p), ("nk_dev", C.c_void_p),
                ("win_grid", C.c_int32), ("win_nw", C.c_int32), ("pad_q", C.c_void_p), ("pad_k", C.c_void_p), ("pad_v", C.c_void_p)]


class AttnBwdArgs(_Sized):
    _fields_ = ([("struct_size", C.c_uint32), ("reserved0", C.c_uint32)] + [(n, C.c_void_p) for n in ("Q", "K", "V", "O", "dO", "dQ", "dK", "dV")] +
                [(f"{t}_stride_{s}", C.c_int64) for t in ("q", "k", "v", "o", "do", "dq", "dk", "dv") for s in ("b", "h", "row")] +
                [("batch", C.c_int32), ("heads", C.c_int32), ("Nq", C.c_int32), ("Nk", C.c_int32), ("head_dim", C.c_int32),
                 ("scale", C.c_float), ("causal", C.c_int32), ("key_mask", C.c_void_p), ("lse", C.c_void_p), ("delta", C.c_void_p),
                 ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p), ("delta_ready", C.c_int32), ("reserved1", C.c_int32)])


class Dropout(C.Structure):
    _fields_ = [("rng_state", C.c_void_p), ("stream", C.c_uint32), ("drop_thr", C.c_uint32), ("seg_rows", C.c_uint32), ("reserved0", C.c_uint32)]


_i64, _i32, _f32, _p = C.c_int64, C.c_int32, C.c_float, C.c_void_p
_dp = C.POINTER(Dropout)

# name -> argtypes (restype is int unless noted); must list EVERY symbol of include/llmseg_hip.h
SIGNATURES = {
    "llmseg_version": [],
    "llmseg_struct_size": [C.c_int],
    "llmseg_launch_count": [],
    "llmseg_last_error": [],
    "llmseg_gemm_bf16": [C.POINTER(GemmArgs), _p],
    "llmseg_gemm_set_variant": [C.c_int],
    "llmseg_attn_fwd": [C.POINTER(AttnArgs), _p],
    "llmseg_attn_set_variant": [C.c_int],
    "llmseg_attn_bwd": [C.POINTER(AttnBwdArgs), _p],
    "llmseg_norm": [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _f32, C.c_int, _p, _p],
    "llmseg_rope": [_p, _p, _p, _i64, _i64, _i32, _i32, _i64, _p],
    "llmseg_rope_kv_append": [_p, _i64, _p, _p, _p, _p, _i64, _p, _i64, _i32, _i32, _p],
    "llmseg_decode_attn": [_p, _i64, _p, _p, _p, _p, _i64, _p, _i64, _i32, _i32, _f32, _p, _i64, _p, _i64, _p],
    "llmseg_swiglu": [_p, _p, _i64, _i64, _i64, _i64, _p],
    "llmseg_act": [_p, _p, _i64, _i32, _p],
    "llmseg_sam_postprocess": [_p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p],
    "llmseg_sam_mask_stats": [_p, _p, _f32, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _p],
    "llmseg_sam_binarize": [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _p],
    "llmseg_nms": [_p, _p, _i32, _f32, _p, _p],
    "llmseg_image_resize_workspace": [_i32, _i32, _i32, _i32, _i32],
    "llmseg_image_resize_u8": [_p, _i64, _p, _i32, _i32, _i32, _i32, _i32, _p, _i64, _p],
    "llmseg_sam_preprocess": [_p, _p, _i32, _i32, _i32, _p, _p, _p],
    "llmseg_mask_small_regions_workspace": [_i32, _i32, _i32],
    "llmseg_mask_small_regions": [_p, _i32, _i32, _i32, _i32, _p, _p, _i64, _p],
    "llmseg_mask_boxes": [_p, _i32, _i32, _i32, _p, _p, _p, _i64, _p],
    "llmseg_add_rows": [_p, _p, _p, _i64, _i64, _i64, _p],
    "llmseg_patchify": [_p, _p, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _p],
    "llmseg_im2col3x3": [_p, _p, _i32, _i32, _i32, _i32, _p],
    "llmseg_embed_splice": [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _i64, _i64, _p],
    "llmseg_gather_rows": [_p, _p, _p, _i64, _i64, _i64, _p],
    "llmseg_mask_pullback": [_p, _p, _p, _p, _i32, _i32, _i32, _p],
    "llmseg_upsample_maskpool": [_p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _p],
    "llmseg_cosine_scores": [_p, _p, _p, _i32, _i32, _p],
    "llmseg_linear_f32": [_p, _i64, _p, _i64, _i32, _p, _p, _i64, _p, _i64, _i32, _i32, _i32, _i32, _f32, _p],
    "llmseg_layernorm_f32": [_p, _p, _p, _p, _i64, _i32, _f32, _p],
    "llmseg_attn_f32": [_p, _p, _p, _p, C.POINTER(C.c_int64), _i32, _i32, _i32, _i32, _i32, _f32, _p],
    "llmseg_cosine_f32": [_p, _p, _p, _i32, _i32, _p],
    "llmseg_align_reg_loss": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _f32, _i32, _p],
    "llmseg_dice_bce": [_p, _p, _p, _i32, _i64, _f32, _p, _i64, _p],
    "llmseg_dice_bce_bwd": [_p, _p, _p, _p, _i32, _i64, _f32, _p],
    "llmseg_ce_loss": [_p, _p, _p, _i32, _i32, _i64, _i64, _p, _i64, _p],
    "llmseg_intersection_union": [_p, _p, _i64, _i32, _p, _p],
    "llmseg_union_resize_iou": [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p],
    "llmseg_rle_decode": [_p, _p, _p, _i32, _i32, _i32, _i32, _p],
    "llmseg_mask_targets": [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p],
    "llmseg_gt_resample": [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _p],
    "llmseg_proposal_targets": [_p, _p, _p, _i32, _p, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _i32, _p, _p, _p, _p, _p],
    "llmseg_resize_aa": [_p, _p, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _i32, _p],
    "llmseg_colsum": [_p, _p, _i64, _i64, _i64, _p, _i64, _p],
    "llmseg_norm_bwd": [_p, _p, _p, _p, _p, _p, _i64, _i64, _f32, C.c_int, _p, _i64, _p],
    "llmseg_norm_bwd_add": [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _f32, C.c_int, _p, _i64, _p],
    "llmseg_swiglu_bwd": [_p, _p, _p, _i64, _i64, _p],
    "llmseg_act_bwd": [_p, _p, _p, _i64, C.c_int, _p],
    "llmseg_softmax_rows": [_p, _p, _i64, _i32, _i32, _i32, _f32, _i32, _p, _i32, _p],
    "llmseg_attn_ds": [_p, _p, _p, _i64, _i32, _i32, _f32, _p],
    "llmseg_ce_bwd": [_p, _p, _p, _p, _i32, _i32, _i64, _i64, _p],
    "llmseg_scatter_add_rows": [_p, _p, _p, _i64, _i64, _p],
    "llmseg_lora_down": [_p, _p, _i64, _p, _p, _p, _i64, _i64, _i64, _i32, _f32, _i32, _dp, _p],
    "llmseg_lora_down_ws": [_p, _p, _i64, _p, _p, _p, _i64, _i64, _i64, _i32, _f32, _i32, _dp, _p, _i64, _p],
    "llmseg_lora_outer": [_p, _p, _i64, _p, _p, _i64, _p, _p, _i64, _i64, _i32, _f32, _dp, _p, _i64, _p],
    "llmseg_lora_wgrads": [_p, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _p, _p, _i64, _i64, _f32, _dp, _p, _i64, _p],
    "llmseg_lora_apply": [_p, _i64, _p, _i64, _p, _p, _i64, _i64, _i32, _f32, _dp, _p],
    "llmseg_lora_pack": [_p, _p, _p, _p, _p, _p, _p, _i64, _f32, _p],
    "llmseg_lora_down_parts": [_p, _p, _i64, _p, _p, _p, _i64, _i64, _i64, _i32, _f32, _i32, _dp, _p, _i64, C.POINTER(C.c_int32), C.POINTER(C.c_float), _p],
    "llmseg_lora_down_pack": [_p, _p, _i64, _p, _p, _p, _i64, _i64, _i64, _i32, _f32, _i32, _dp, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _i64, _f32, _p],
    "llmseg_transpose_pad": [_p, _p, _i64, _i64, _i64, _i64, _i64, _p],
    "llmseg_sumsq": [_p, _i64, C.c_int, _p, _p, _i64, _p],
    "llmseg_adamw": [_p, _p, _p, C.c_int, _p, _p, _i64, _f32, _f32, _f32, _f32, _f32, _i64, _p, _p],
    "llmseg_prof_enable": [C.c_int],
    "llmseg_prof_dominant_kernel": [],
    "llmseg_prof_dominant_bytes": [],
    "llmseg_prof_dominant_info": [C.POINTER(C.c_int64)],
    "llmseg_prof_collect": [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double),
                            C.POINTER(C.c_int64)],
}

_lib = None


def load():
    """Load the shared library and bind every declared symbol (raises if anything is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: the HIP extension is not built "
                           f"(run llmseg_amd/csrc/build.sh); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = (C.c_char_p if name in ("llmseg_last_error", "llmseg_prof_dominant_kernel") else
                      C.c_double if name == "llmseg_prof_dominant_bytes" else C.c_int64 if name in ("llmseg_struct_size", "llmseg_launch_count", "llmseg_image_resize_workspace", "llmseg_mask_small_regions_workspace") else C.c_int)
    # ABI guard at load time: this binding's structs must be the library's (the entry points check `struct_size` per call as well)
    if lib.llmseg_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH}: ABI version {lib.llmseg_version()} != {ABI_VERSION} of this binding (rebuild: llmseg_amd/csrc/build.sh)")
    for which, st in enumerate((GemmArgs, AttnArgs, AttnBwdArgs, Dropout)):
        if lib.llmseg_struct_size(which) != C.sizeof(st):
            raise RuntimeError(f"{LIB_PATH}: sizeof({st.__name__}) = {C.sizeof(st)} here, {lib.llmseg_struct_size(which)} in the library")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {load().llmseg_last_error().decode()}")
