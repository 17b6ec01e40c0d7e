"""ctypes binding of the C ABI in include/idiff.h (instancediff_amd/libidiff_hip.so).

The library is the product: there is no CPU / ATen fallback.  If it is missing or a symbol declared in
include/idiff.h is not exported, importing the compute path fails loudly.
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IDIFF_LIB") or os.path.join(_HERE, "libidiff_hip.so")  # IDIFF_LIB: A/B kernel experiments
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "idiff.h")

c_f32p = C.c_void_p  # device pointers travel as integers
c_stream = C.c_void_p


class ConvDesc(C.Structure):
    _fields_ = [
        ("src0", C.c_void_p), ("src1", C.c_void_p),
        ("src0_bstride", C.c_int64), ("src1_bstride", C.c_int64),
        ("C0", C.c_int32), ("C1", C.c_int32),
        ("B", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32),
        ("mode", C.c_int32), ("ks", C.c_int32), ("Cout", C.c_int32),
        ("wpk", C.c_void_p), ("bias", C.c_void_p),
        ("pro_a", C.c_void_p), ("pro_b", C.c_void_p),
        ("out", C.c_void_p), ("out_bstride", C.c_int64),
        ("res", C.c_void_p), ("res_bstride", C.c_int64),
        ("vec", C.c_void_p),
        ("aux", C.c_void_p), ("aux_bstride", C.c_int64), ("aux_a", C.c_void_p), ("aux_b", C.c_void_p),
        ("stats", C.c_void_p),
        ("wwino", C.c_void_p),
        ("wwino4", C.c_void_p),
        ("gn_gamma", C.c_void_p), ("gn_beta", C.c_void_p), ("gn_film", C.c_void_p), ("gn_film_ld", C.c_int64), ("gn_eps", C.c_float),
        ("gn_groups", C.c_int32), ("gn_out_a", C.c_void_p), ("gn_out_b", C.c_void_p), ("gn_mean_rstd", C.c_void_p),
        ("algo_request", C.c_int32),
        ("wx3", C.c_void_p),
    ]


class LinearGroup(C.Structure):  # idiff_linear_group
    _fields_ = [("x", C.c_void_p), ("ldx", C.c_int64), ("wT", C.c_void_p), ("ldw", C.c_int64), ("bias", C.c_void_p), ("res", C.c_void_p),
                ("ldr", C.c_int64), ("gscale", C.c_void_p), ("out", C.c_void_p), ("ldo", C.c_int64), ("ln_g", C.c_void_p), ("ln_b", C.c_void_p),
                ("ln_eps", C.c_float), ("R", C.c_int32), ("K", C.c_int32), ("N", C.c_int32), ("act_in", C.c_int32), ("act_out", C.c_int32)]


class XattnGroup(C.Structure):  # idiff_xattn_group
    _fields_ = [("qf", C.c_void_p), ("mem", C.c_void_p), ("o", C.c_void_p), ("ws", C.c_void_p), ("Cm", C.c_int32), ("N", C.c_int32)]


class MemprojGroup(C.Structure):  # idiff_memproj_group
    _fields_ = [("feat", C.c_void_p), ("feat_bstride", C.c_int64), ("ln1_g", C.c_void_p), ("ln1_b", C.c_void_p), ("gram", C.c_void_p),
                ("hvec", C.c_void_p), ("evar", C.c_float), ("out", C.c_void_p), ("C", C.c_int32), ("N", C.c_int32), ("Cm", C.c_int32)]


class ScoremapGroup(C.Structure):  # idiff_scoremap_group
    _fields_ = [("feat", C.c_void_p), ("feat_bstride", C.c_int64), ("tv", C.c_void_p), ("out", C.c_void_p), ("sel", C.c_void_p),
                ("C", C.c_int32), ("HW", C.c_int32)]


LINEAR_MAX_GROUPS = 16
XATTN_MAX_GROUPS, MEMPROJ_MAX_GROUPS, SCOREMAP_MAX_GROUPS = 8, 4, 4
P, I, I64, F, U64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64

# name -> (restype, argtypes); must list every function declared in include/idiff.h
SIGNATURES = {
    "idiff_last_error": (C.c_char_p, []),
    "idiff_version": (I, []),
    "idiff_launch_count": (I64, []),
    "idiff_device_info": (I, [C.POINTER(I), C.POINTER(I), C.c_char_p, I]),
    "idiff_conv2d_num_tiles": (I, [I, I]),
    "idiff_conv2d_fwd": (I, [C.POINTER(ConvDesc), c_stream]),
    "idiff_conv2d_last_algo": (I, []),
    "idiff_conv2d_plan": (I, [C.POINTER(ConvDesc)]),
    "idiff_pack_conv_weight": (I, [P, P, I, I, I, c_stream]),
    "idiff_pack_conv_weight_T": (I, [P, P, I, I, I, c_stream]),
    "idiff_pack_conv_weight_wino": (I, [P, P, I, I, I, c_stream]),
    "idiff_pack_conv_weight_wino4": (I, [P, P, I, I, I, c_stream]),
    "idiff_conv1x1_x3_image_bytes": (C.c_longlong, [I, I]),
    "idiff_pack_conv1x1_x3": (I, [P, P, I, I, c_stream]),
    "idiff_gn_finalize": (I, [P, I, I, I, I, I, P, P, P, I64, F, P, P, P, c_stream]),
    "idiff_affine_silu_add": (I, [P, I64, P, P, P, I64, P, P, I64, I, I, I, c_stream]),
    "idiff_linear_fwd": (I, [P, I64, P, I64, P, P, I64, P, P, I64, I, I, I, I, I, c_stream]),
    "idiff_linear_t_fwd": (I, [P, I64, P, I64, P, P, I64, P, P, I64, I, I, I, I, I, c_stream]),
    "idiff_linear_t_ln_fwd": (I, [P, I64, P, P, F, P, I64, P, P, I64, P, P, I64, I, I, I, I, c_stream]),
    "idiff_linear_t_grouped_fwd": (I, [C.POINTER(LinearGroup), I, c_stream]),
    "idiff_attn_tokens_grouped_fwd": (I, [C.POINTER(P), C.POINTER(P), C.POINTER(P), C.POINTER(P), I, I, I, I, I, I, F, I64, I64, c_stream]),
    "idiff_linear_t_heads_fwd": (I, [P, I64, I64, P, I64, I64, P, I64, P, I64, I64, I, I, I, I, c_stream]),
    "idiff_smm_memproj_fwd": (I, [P, I64, P, P, P, P, P, P, P, I, I, I, F, c_stream]),
    "idiff_smm_memproj_compact_fwd": (I, [P, I64, P, P, P, P, F, P, I, I, I, I, F, F, c_stream]),
    "idiff_smm_memproj_compact_train_fwd": (I, [P, I64, P, P, P, P, P, P, I, I, I, I, F, F, c_stream]),
    "idiff_smm_memproj_compact_bwd_ws_floats": (I64, [I, I, I]),
    "idiff_smm_memproj_compact_bwd": (I, [P, I64, P, P, P, P, P, P, P, I64, P, P, I, I, I, I, F, F, c_stream]),
    "idiff_smm_memproj_compact_grouped_fwd": (I, [C.POINTER(MemprojGroup), I, I, F, F, c_stream]),
    "idiff_layernorm_rows_fwd": (I, [P, I64, P, P, P, I64, I, I, F, P, c_stream]),
    "idiff_time_embed_fwd": (I, [P, P, I, I, P, c_stream]),
    "idiff_time_mlp_fwd": (I, [P, P, P, P, P, P, P, I, I, I, I, c_stream]),
    "idiff_chan_layernorm_fwd": (I, [P, I64, P, P, P, I64, I, I, I, F, P, c_stream]),
    "idiff_attn_self_fwd": (I, [P, P, P, I, I, I, I, F, c_stream]),
    "idiff_attn_self_bf16_fwd": (I, [P, P, I, I, I, I, F, c_stream]),
    "idiff_attn_self_f16_fwd": (I, [P, P, I, I, I, I, F, c_stream]),
    "idiff_attn_ctx_fwd": (I, [P, P, P, P, I, I, I, I, I, F, c_stream]),
    "idiff_attn_tokens_fwd": (I, [P, P, P, P, I, I, I, I, I, F, I64, I64, c_stream]),
    "idiff_attn_tokens_bwd": (I, [P, P, P, P, P, P, P, I, I, I, I, I, F, I64, I64, I64, I64, I64, c_stream]),
    "idiff_smm_xattn_ws_floats": (I64, [I, I, I, I, I]),
    "idiff_attn_tokens_f16_fwd": (I, [P, P, P, P, I, I, I, I, I, F, I64, I64, c_stream]),
    "idiff_smm_xattn_kv_f16_ws_floats": (I64, [I, I]),
    "idiff_smm_xattn_kv_f16_fwd": (I, [P, P, P, P, P, I, I, I, I, I, F, c_stream]),
    "idiff_smm_xattn_fwd": (I, [P, P, P, P, I, I, I, I, I, F, c_stream]),
    "idiff_smm_xattn_grouped_fwd": (I, [C.POINTER(XattnGroup), I, I, I, I, F, c_stream]),
    "idiff_smm_xattn_lse_fwd": (I, [P, P, P, P, P, I, I, I, F, c_stream]),
    "idiff_smm_xattn_bwd": (I, [P, P, P, P, P, P, P, I, P, I, I, I, F, c_stream]),
    "idiff_smm_xattn_cm_lse_fwd": (I, [P, P, P, P, P, I, I, I, I, F, c_stream]),
    "idiff_smm_xattn_cm_bwd": (I, [P, P, P, P, P, P, P, I, P, I, I, I, I, F, c_stream]),
    "idiff_scoremap_fwd": (I, [P, I64, P, P, P, P, I, I, I, I, c_stream]),
    "idiff_scoremap_grouped_fwd": (I, [C.POINTER(ScoremapGroup), I, P, I, I, c_stream]),
    "idiff_gather_channel": (I, [P, P, P, I, I, I, c_stream]),
    "idiff_conv3x3_select_fwd": (I, [P, I64, P, P, P, P, I, I, I, I, I, c_stream]),
    "idiff_conv3x3_select_bwd_ws_floats": (I64, [I, I, I, I]),
    "idiff_conv3x3_select_bwd": (I, [P, I64, P, P, P, P, I64, P, P, P, I, I, I, I, I, c_stream]),
    "idiff_irsde_reverse_step": (I, [P, P, P, P, P, I64, F, F, F, F, F, I, U64, U64, c_stream]),
    "idiff_irsde_map": (I, [I, P, P, P, P, F, P, I, I64, P, C.POINTER(C.c_float), U64, U64, c_stream]),
    "idiff_drift_reverse_step": (I, [P, P, P, P, P, P, P, I64, F, F, F, U64, U64, c_stream]),
    "idiff_drift_reverse_step_dev": (I, [P, P, P, P, P, P, I64, P, I, P, U64, U64, U64, c_stream]),
    "idiff_step_state_advance": (I, [P, P, I, I, I, c_stream]),
    "idiff_randn": (I, [P, I64, U64, U64, c_stream]),
    "idiff_dropout": (I, [P, P, I64, F, U64, U64, c_stream]),
    "idiff_philox_raw": (I, [P, I64, U64, U64, c_stream]),
    "idiff_axpby": (I, [P, P, P, I64, F, F, c_stream]),
    "idiff_mix3_per_sample": (I, [P, P, P, P, P, P, P, I, I64, c_stream]),
    # ---- training path ----
    "idiff_conv2d_wgrad_ws_floats": (I64, [C.POINTER(ConvDesc)]),
    "idiff_conv2d_wgrad": (I, [C.POINTER(ConvDesc), P, I64, P, I, P, c_stream]),
    "idiff_conv2d_wgrad_last_algo": (I, []),
    "idiff_sumpool2x2": (I, [P, P, I64, I, I, c_stream]),
    "idiff_pixel_shuffle2": (I, [P, P, I, I, I, I, c_stream]),
    "idiff_plane_sum": (I, [P, I64, P, I, I, I, c_stream]),
    "idiff_batch_sum": (I, [P, P, I, I, I, c_stream]),
    "idiff_sum_n": (I, [C.POINTER(P), C.POINTER(I64), I, P, I64, I, I64, c_stream]),
    "idiff_gather_segments": (I, [P, I, I64, P, c_stream]),
    "idiff_gn_silu_bwd_ws_floats": (I64, [I, I, I]),
    "idiff_gn_silu_bwd": (I, [P, I64, P, I64, P, P, P, P, P, P, I64, P, I64, P, P, P, I64, P, I, I, I, I, I, P, P, c_stream]),
    "idiff_act_fwd": (I, [P, P, I64, I, c_stream]),
    "idiff_act_bwd": (I, [P, P, P, I64, I, c_stream]),
    "idiff_colsum": (I, [P, I64, P, I, I, I, c_stream]),
    "idiff_scale_cols": (I, [P, P, P, I, I, c_stream]),
    "idiff_colsum_prod": (I, [P, P, P, I, I, c_stream]),
    "idiff_layernorm_rows_bwd": (I, [P, I64, P, I64, P, P, P, I64, P, P, I, I, I, c_stream]),
    "idiff_chan_layernorm_bwd_ws_floats": (I64, [I, I, I]),
    "idiff_chan_layernorm_bwd": (I, [P, I64, P, I64, P, P, P, I64, P, P, P, I, I, I, I, c_stream]),
    "idiff_chan_normalize_fwd": (I, [P, I64, P, P, I, I, I, c_stream]),
    "idiff_chan_normalize_bwd": (I, [P, P, P, P, I64, I, I, I, c_stream]),
    "idiff_scatter_channel": (I, [P, P, P, I, I, I, c_stream]),
    "idiff_bgemm_ws_floats": (I64, [I, I, I, I]),
    "idiff_bgemm": (I, [P, P, P, I, I, I, I64, I64, I64, I, I, I64, I64, I64, I, F, F, P, c_stream]),
    "idiff_bgemm_bias": (I, [P, P, P, I, I, I, I64, I64, I64, I, I, I64, I64, I64, I, F, P, I64, c_stream]),
    "idiff_layernorm_rows_g_fwd": (I, [P, I64, P, P, P, I64, I, I, F, P, I, c_stream]),
    "idiff_layernorm_rows_g_bwd": (I, [P, I64, P, I64, P, P, P, I64, P, P, I, I, I, c_stream]),
    "idiff_colsum_g": (I, [P, I64, P, I, I, I, c_stream]),
    "idiff_linear_mfma_fwd": (I, [P, I64, P, I64, P, P, I64, I, I, I, c_stream]),
    "idiff_softmax_rows_fwd": (I, [P, I64, P, I64, I, I, F, c_stream]),
    "idiff_softmax_rows_bwd": (I, [P, I64, P, I64, P, I64, I, I, F, c_stream]),
    "idiff_resize_bilinear": (I, [P, P, I64, I, I, I, I, c_stream]),
    "idiff_mse_loss": (I, [P, P, P, P, P, I64, F, c_stream]),
    "idiff_image_metrics": (I, [P, P, P, P, I, I, I, c_stream]),
    "idiff_f32_to_bf16": (I, [P, P, I64, c_stream]),
    "idiff_bf16_to_f32": (I, [P, P, I64, c_stream]),
    "idiff_adam_step": (I, [P, P, P, P, I64, F, F, F, F, F, F, I, c_stream]),
}


def header_symbols(path=HEADER_PATH):
    """Every function name declared in include/idiff.h (used by the CPU test that the library exports them all)."""
    with open(path) as f:
        txt = f.read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(idiff_[A-Za-z0-9_]+)\s*\(", txt)))


class IdiffError(RuntimeError):
    pass


_lib = None


def load():
    """Load the HIP kernel library (idempotent).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IdiffError(
            f"{LIB_PATH} not found: the HIP kernel library has not been built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `make -C instancediff_amd/csrc`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise IdiffError(f"libidiff_hip.so does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().idiff_last_error()
        raise IdiffError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
