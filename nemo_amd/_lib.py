"""ctypes binding of libmi355x_asr.so (the C ABI declared in include/mi355x_asr.h).

The library is the product: there is NO fallback.  If it is missing or an entry point is absent, importing
this module raises, and every wrapper in `nemo_amd.ops` fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

# torch FIRST: the ROCm wheel carries its own libamdhip64 (torch/lib); a process must run ONE HIP runtime, and it has to be the
# one torch's streams / allocator / events live in.  Loaded before torch, this library would pull /opt/rocm's copy in under the
# same soname and torch would silently run on it (seen as hipFuncSetAttribute failing on the first launch).
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355X_ASR_LIB", os.path.join(_HERE, "lib", "libmi355x_asr.so"))

vp, i32, i64, f32, f64, u32 = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_double, C.c_uint


class GemmDesc(C.Structure):
    """mirror of `mi355x_gemm_desc` (include/mi355x_asr.h)"""
    _fields_ = [
        ("A", vp), ("B", vp), ("C", vp),
        ("M", i32), ("N", i32), ("K", i32),
        ("lda", i64), ("ldb", i64), ("ldc", i64), ("c_col_stride", i64),
        ("transA", i32), ("transB", i32),
        ("in_dtype", i32), ("c_dtype", i32),
        ("batch", i32), ("nb0", i32),
        ("sA0", i64), ("sA1", i64), ("sB0", i64), ("sB1", i64), ("sC0", i64), ("sC1", i64),
        ("bias", vp), ("alpha", f32),
        ("epilogue", i32), ("atomic", i32), ("splitk", i32),
        ("aux_in", vp), ("aux_in_dtype", i32),
        ("aux_out", vp), ("aux_out_dtype", i32),
        ("ldaux", i64),
        ("drop_key", u32), ("drop_threshold", u32), ("drop_scale", f32),
        ("row_len", vp), ("rows_per_b", i32), ("rows_inner", i32),
        ("colsum_stride", i64), ("colsum_out", vp),
        ("gather", vp), ("rowmap", vp),
    ]


class ConvGather(C.Structure):  # mirrors mi355x_conv_gather
    _fields_ = [("nI", i32), ("nJ", i32), ("SI", i32), ("SJ", i32), ("C", i32), ("si", i32), ("sj", i32), ("ntaps", i32),
                ("di", i32 * 9), ("dj", i32 * 9), ("operand", i32)]


class RowMap(C.Structure):  # mirrors mi355x_row_map
    _fields_ = [("nI", i32), ("nJ", i32), ("OI", i32), ("OJ", i32), ("si", i32), ("sj", i32), ("oi", i32), ("oj", i32)]


class FfnPackEntry(C.Structure):
    """mirror of `mi355x_ffn_pack_entry`"""
    _fields_ = [("src", vp), ("k512", vp), ("kchunk", vp), ("d_ff", i32), ("is_w2", i32)]


class PackEntry(C.Structure):
    """mirror of `mi355x_pack_entry`"""
    _fields_ = [
        ("src", vp), ("dst", vp),
        ("rows", i32), ("cols", i32), ("nr2", i32), ("nc2", i32),
        ("sr1", i64), ("sr2", i64), ("sc1", i64), ("sc2", i64), ("pitch", i64),
        ("tile_begin", i64),
    ]


# name -> argtypes (return type is always int, except the version string)
SIGNATURES = {
    "mi355x_gemm": [C.POINTER(GemmDesc), vp],
    "mi355x_gemm_grouped": [C.POINTER(GemmDesc), i32, vp],
    "mi355x_gemm_config": [i32, i32],
    "mi355x_set_step_counter": [vp],
    "mi355x_set_null_launch": [i32],
    "mi355x_ffn_fwd": [vp, i64, vp, vp, vp, vp, vp, i64, vp, i64, vp, i64, i32, i32, i32, f32, u32, u32, f32, u32, u32, f32, vp],
    "mi355x_ffn_pack": [vp, i32, i32, vp],
    "mi355x_ffn_bwd_dgrad": [vp, i64, vp, vp, vp, i64, vp, vp, vp, i64, i32, i32, i32, u32, u32, f32, vp],
    "mi355x_logmel_fwd": [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, i32, f32, f32, u32, f32, vp, i32, i32, i32, vp],
    "mi355x_feat_normalize": [vp, vp, vp, i32, i32, i32, i32, i32, f32, vp],
    "mi355x_subsample_conv1_fwd": [vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, vp],
    "mi355x_subsample_conv1_bwd": [vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp, i64, vp],
    "mi355x_im2col_3x3s2": [vp, vp, i32, i32, i32, i32, i32, vp],
    "mi355x_col2im_3x3s2_relu": [vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "mi355x_dwconv2d_s2_fwd": [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "mi355x_dwconv2d_s2_bwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, i64, vp],
    "mi355x_subsample_conv1_fwd_pad": [vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp],
    "mi355x_subsample_conv1_bwd_pad": [vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, i64, vp],
    "mi355x_im2col_3x3s2_pad": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "mi355x_col2im_3x3s2_relu_pad": [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "mi355x_dwconv2d_s2_fwd_pad": [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "mi355x_dwconv2d_s2_bwd_pad": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, i64, vp],
    "mi355x_embed_sos_fwd": [vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "mi355x_embed_sos_bwd": [vp, vp, i32, vp, i32, i32, i32, i32, vp],
    "mi355x_lstm_cell_fwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "mi355x_lstm_cell_bwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "mi355x_joint_combine_fwd": [vp, vp, vp, i32, u32, u32, f32, i32, i32, i32, i32, vp],
    "mi355x_joint_combine_bwd": [vp, vp, vp, i32, f32, i32, i32, i32, i32, vp],
    "mi355x_cast_rows": [vp, i64, vp, i32, i64, i64, i32, i32, f32, vp],
    "mi355x_layernorm_fwd": [vp, i32, vp, vp, vp, i32, vp, vp, i32, i32, f32, vp],
    "mi355x_layernorm2_fwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, f32, vp],
    "mi355x_layernorm2_bwd": [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, f32, u32, u32, f32, vp],
    "mi355x_layernorm_bwd": [vp, i32, vp, i32, vp, vp, vp, vp, i32, vp, vp, i32, i32, vp],
    "mi355x_layernorm_bwd_cast": [vp, i32, vp, i32, vp, vp, vp, vp, i32, vp, vp, i32, i32, vp, f32, u32, u32, f32, vp],
    "mi355x_colsum": [vp, i32, i64, vp, i32, i32, f32, vp],
    "mi355x_log_softmax_fwd": [vp, i64, vp, i64, i32, i32, vp],
    "mi355x_log_softmax_bwd": [vp, vp, i64, vp, i32, i64, i32, i32, f32, vp],
    "mi355x_scale_bias_fwd": [vp, vp, vp, vp, i32, i64, i32, i32, vp],
    "mi355x_scale_bias_bwd": [vp, i32, i32, vp, vp, vp, vp, vp, i64, i32, vp, i64, vp],
    "mi355x_cast_pitched": [vp, vp, i32, i64, i32, i32, f32, u32, u32, f32, vp],
    "mi355x_swish_mask_fwd": [vp, vp, i32, vp, i32, i64, i32, vp],
    "mi355x_swish_mask_bwd": [vp, vp, vp, i32, vp, i32, i64, i32, vp],
    "mi355x_time_reduce_dwconv_fwd": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "mi355x_time_reduce_dwconv_bwd": [vp, i32, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "mi355x_time_recover_fwd": [vp, vp, vp, i32, i32, i32, vp],
    "mi355x_time_recover_bwd": [vp, vp, i32, i32, i32, i32, i32, vp],
    "mi355x_glu_fwd": [vp, vp, i32, vp, i32, i64, i32, vp, vp],
    "mi355x_glu_bwd": [vp, vp, vp, i32, vp, i32, i64, i32, vp, vp],
    "mi355x_rows_pack": [vp, vp, i32, i64, i64, vp, vp, i32, i64, i32, i32, vp],
    "mi355x_drop_scale_cast": [vp, i32, vp, i32, i64, f32, u32, u32, f32, vp],
    "mi355x_qbias": [vp, i64, vp, vp, vp, vp, i32, i64, i32, vp],
    "mi355x_add2": [vp, vp, i32, vp, i32, i64, i64, i32, vp],
    "mi355x_ctc_greedy_decode": [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "mi355x_fill_rects": [vp, vp, i32, i32, i32, i32, f32, vp],
    "mi355x_specaug_rects": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, i32, vp],
    "mi355x_add2_colsum": [vp, vp, vp, i64, i64, i32, vp, vp, i64, vp],
    "mi355x_relpos_softmax_fwd": [vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, f32, u32, u32, f32, vp],
    "mi355x_relpos_softmax_fwd_ctx": [vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, f32, u32, u32, f32, i32, i32, i32, vp],
    "mi355x_relpos_softmax_bwd": [vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, u32, u32, f32, vp],
    "mi355x_relpos_flash_fwd": [vp, i64, vp, i64, vp, vp, vp, vp, vp, i64, vp, i32, i32, i32, i32, i32, f32, u32, u32, f32, vp, vp],
    "mi355x_attn_delta": [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp],
    "mi355x_attn_bwd_prep": [vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp],
    "mi355x_relpos_flash_bwd_dq": [vp, vp, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, i64, i32, i32, i32, i32, i64,
                                   f32, u32, u32, f32, vp, vp],
    "mi355x_relpos_flash_bwd_dkv": [vp, vp, vp, i64, vp, i64, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, f32, u32, u32, f32, vp, vp],
    "mi355x_relpos_flash_bwd_dpos": [vp, vp, vp, vp, i64, vp, vp, i64, i32, i32, i32, i32, i64, vp, vp],
    "mi355x_relpos_ds_elems": [i32, i32, i32],
    "mi355x_relpos_dpos_partial_elems": [i32, i32, i32],
    "mi355x_dwconv_fwd": [vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, vp],
    "mi355x_dwconv_fwd_ctx": [vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, vp],
    "mi355x_dwconv_fwd_glu": [vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, vp],
    "mi355x_dwconv_bwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, i64, vp],
    "mi355x_dwconv_bwd_ctx": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, i64, vp],
    "mi355x_dwconv_bwd_bnswish": [vp, vp, vp, vp, vp, vp, vp, f64, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, i64, i32, vp],
    "mi355x_dwconv_tap_reduce": [vp, i64, i32, i32, i32, vp, vp, vp],
    "mi355x_bn_finalize": [vp, f64, vp, vp, vp, vp, f32, f32, i32, vp],
    "mi355x_bn_finalize_dev_count": [vp, vp, vp, vp, vp, vp, f32, f32, i32, vp],
    "mi355x_bn_eval_stats": [vp, vp, vp, vp, f32, i32, vp],
    "mi355x_bn_swish_fwd": [vp, vp, vp, vp, vp, vp, i32, i64, i32, vp],
    "mi355x_bn_swish_bwd_reduce": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i64, i32, vp, i64, vp],
    "mi355x_bn_stats_swish_fwd": [vp, vp, f64, vp, vp, vp, vp, vp, vp, vp, vp, f32, f32, i32, i64, i32, vp],
    "mi355x_bn_swish_bwd_apply": [vp, vp, vp, vp, vp, vp, vp, f64, i32, vp, i32, i64, i32, vp],
    "mi355x_bn_swish_bwd_apply_dev_count": [vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, i64, i32, vp],
    "mi355x_bn_param_grad": [vp, vp, vp, i32, vp],
    "mi355x_ctc_loss": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, vp],
    "mi355x_row_scale": [vp, vp, i64, i64, vp],
    "mi355x_adamw_step": [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, f32, vp],
    "mi355x_adamw_step_ex": [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, f32, vp, vp, f32, vp],
    "mi355x_grad_sumsq": [vp, i64, vp, vp],
    "mi355x_clip_coef": [vp, i32, f32, f32, vp, vp],
    "mi355x_pack_weights": [vp, i32, i64, i32, vp],
    "mi355x_fill_f32": [vp, i64, f32, vp],
    "mi355x_rnnt_workspace_elems": [i32, i32, i32, vp],
    "mi355x_rnnt_loss": [vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, f32, f32, vp, vp, vp, i64, vp],
    "mi355x_rnnt_greedy_decode": [vp, i32, i64, vp, vp, vp, i64, vp, i64, vp, vp, vp, i64, vp, vp, i64, vp, i32, i32, i32, i32, i32,
                                  i32, i32, i32, vp, vp, vp, vp, i32, vp, vp, vp],
    "mi355x_dwconv_config": [i32],
    "mi355x_ctc_config": [i32],
    "mi355x_logmel_config": [i32],
    "mi355x_stream_create": [i32, vp],
    "mi355x_stream_destroy": [vp],
    "mi355x_tape_log_begin": [vp],
    "mi355x_tape_log_end": [],
    "mi355x_tape_from_graph": [vp, i32, vp],
    "mi355x_tape_replay": [vp, vp, i32],
    "mi355x_tape_join": [vp, vp],
    "mi355x_tape_info": [vp, vp],
    "mi355x_tape_destroy": [vp],
    "mi355x_mailbox_create": [i32, i32, i32, i32, i32, vp, vp],
    "mi355x_mailbox_open": [vp, i32, vp],
    "mi355x_mailbox_exchange": [vp, vp, i32, vp],
    "mi355x_mailbox_status": [vp, vp],
    "mi355x_mailbox_poll": [vp, vp, vp],
    "mi355x_mailbox_destroy": [vp],
    "mi355x_rnnt_loss_ex": [vp, i64, vp, vp, vp, i32, i32, i32, i32, i32, f32, f32, f32, vp, vp, i32, i64, vp, i64, vp],
}

EXPORTED_SYMBOLS = sorted(list(SIGNATURES) + ["mi355x_asr_version"])


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -m nemo_amd.build` (hipcc --offload-arch=gfx950). "
            "There is no CPU / PyTorch fallback for the MI355X kernels.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing -> loud
        fn.argtypes = argtypes
        fn.restype = i32
    lib.mi355x_asr_version.restype = C.c_char_p
    lib.mi355x_relpos_ds_elems.restype = i64
    lib.mi355x_relpos_dpos_partial_elems.restype = i64
    lib.mi355x_asr_version.argtypes = []
    lib.mi355x_tape_destroy.restype = None
    lib.mi355x_mailbox_destroy.restype = None
    return lib


lib = _load()


def check(rc: int, what: str = ""):
    """0 ok; 1 -> ValueError (the reference's convention for bad shapes/args); 1000+hipError_t -> RuntimeError."""
    if rc == 0:
        return
    if rc == 1:
        raise ValueError(f"libmi355x_asr: invalid argument in {what}")
    if rc == 2:  # MI_ERR_LAUNCH: a launch attribute (dynamic LDS size) was refused before anything was launched
        raise RuntimeError(f"libmi355x_asr: launch configuration refused in {what} (hipFuncSetAttribute failed)")
    raise RuntimeError(f"libmi355x_asr: kernel launch failed in {what} (hipError_t={rc - 1000 if rc >= 1000 else rc})")


def version() -> str:
    return lib.mi355x_asr_version().decode()
