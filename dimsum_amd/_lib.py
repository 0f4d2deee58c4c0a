"""ctypes loader of libdimsum_hip.so + mirrors of the C structs in include/dimsum_hip.h."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DIMSUM_HIP_LIB") or os.path.join(_HERE, "lib", "libdimsum_hip.so")   # override: kernel-variant experiments

F32, F16, BF16 = 0, 1, 2
i32, i64, vp, f32 = C.c_int32, C.c_int64, C.c_void_p, C.c_float


u32 = C.c_uint32


class _Sized(C.Structure):
    """parameter structs of ABI >= 17 start with `struct_size` = sizeof(the struct as the caller knows it): filled in on construction
    (a struct nested inside another one is part of the parent's buffer and is not constructed: the library ignores the nested size)"""

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = C.sizeof(type(self))


class SsmExt(_Sized):
    """dimsum_ssm_ext_t: everything beyond the reference's SSMParamsBase"""
    _fields_ = [("struct_size", u32), ("kernel_variant", i32), ("ckpt_ptr", vp), ("timing_start_event", vp), ("timing_stop_event", vp),
                ("out_z_lo_offset", i64), ("dt_w_ptr", vp), ("dt_x_ptr", vp), ("dt_w_row_stride", i64), ("dt_x_row_stride", i64),
                ("dt_rank", i32), ("out_z_f16", i32), ("out_z_scale_ptr", vp), ("out_z_scale_ld", i64)]


class SsmParams(_Sized):
    _fields_ = ([("struct_size", u32)]
                + [(n, i32) for n in ("batch", "dim", "seqlen", "dstate", "n_groups", "n_chunks", "delta_softplus", "dtype", "reserved")]
                + [(n, i64) for n in ("A_d_stride", "A_dstate_stride", "B_batch_stride", "B_group_stride",
                                      "B_dstate_stride", "C_batch_stride", "C_group_stride", "C_dstate_stride",
                                      "u_batch_stride", "u_d_stride", "delta_batch_stride", "delta_d_stride",
                                      "z_batch_stride", "z_d_stride", "out_batch_stride", "out_d_stride",
                                      "out_z_batch_stride", "out_z_d_stride")]
                + [(n, vp) for n in ("A_ptr", "B_ptr", "C_ptr", "D_ptr", "u_ptr", "delta_ptr", "delta_bias_ptr",
                                     "z_ptr", "out_ptr", "x_ptr", "out_z_ptr")]
                + [("ext", C.POINTER(SsmExt))])


def attach_ext(P, ext_type):
    """a fresh, zeroed extension struct linked to P.ext (P keeps it alive) -> the extension"""
    E = ext_type()
    P.ext = C.pointer(E)
    return E


class SsmBwdParams(_Sized):
    _fields_ = ([("struct_size", u32), ("reserved", u32), ("fwd", SsmParams)]
                + [(n, i64) for n in ("dout_batch_stride", "dout_d_stride", "dA_d_stride", "dA_dstate_stride",
                                      "dB_batch_stride", "dB_group_stride", "dB_dstate_stride", "dC_batch_stride",
                                      "dC_group_stride", "dC_dstate_stride", "du_batch_stride", "du_d_stride",
                                      "dz_batch_stride", "dz_d_stride", "ddelta_batch_stride", "ddelta_d_stride")]
                + [(n, vp) for n in ("dout_ptr", "dA_ptr", "dB_ptr", "dC_ptr", "dD_ptr", "du_ptr", "dz_ptr",
                                     "ddelta_ptr", "ddelta_bias_ptr", "workspace_ptr")]
                + [("workspace_bytes", i64)])


class ConvParams(_Sized):
    _fields_ = ([("struct_size", u32)] + [(n, i32) for n in ("batch", "dim", "seqlen", "width", "silu_activation", "dtype", "reserved")]
                + [(n, i64) for n in ("x_batch_stride", "x_c_stride", "weight_c_stride", "weight_width_stride",
                                      "out_batch_stride", "out_c_stride")]
                + [(n, vp) for n in ("x_ptr", "weight_ptr", "bias_ptr", "out_ptr")])


class ConvBwdParams(_Sized):
    _fields_ = ([("struct_size", u32), ("reserved", u32), ("fwd", ConvParams)]
                + [(n, i64) for n in ("dout_batch_stride", "dout_c_stride", "dx_batch_stride", "dx_c_stride",
                                      "dweight_c_stride", "dweight_width_stride")]
                + [(n, vp) for n in ("dout_ptr", "dx_ptr", "dweight_ptr", "dbias_ptr")])


class NormParams(_Sized):
    _fields_ = ([("struct_size", u32)] + [(n, i32) for n in ("rows", "cols", "is_rms_norm", "x_dtype", "residual_dtype", "out_dtype")]
                + [("eps", f32)]
                + [(n, i64) for n in ("x_row_stride", "residual_row_stride", "y_row_stride", "residual_out_row_stride")]
                + [(n, vp) for n in ("x_ptr", "residual_ptr", "weight_ptr", "bias_ptr", "y_ptr", "residual_out_ptr",
                                     "mean_ptr", "rstd_ptr", "xbias_ptr", "mod_scale_ptr", "mod_shift_ptr")]
                + [("mod_row_stride", i64), ("rows_per_batch", i32), ("y_split3", i32), ("y_inv_scale_ptr", vp)])


class NormBwdParams(_Sized):
    _fields_ = ([("struct_size", u32)] + [(n, i32) for n in ("rows", "cols", "is_rms_norm")] + [("eps", f32), ("reserved", i32)]
                + [(n, i64) for n in ("r_row_stride", "dy_row_stride", "dres_row_stride", "dx_row_stride")]
                + [(n, vp) for n in ("r_ptr", "weight_ptr", "mean_ptr", "rstd_ptr", "dy_ptr", "dres_ptr", "dx_ptr",
                                     "dweight_ptr", "dbias_ptr")])


class TtParams(_Sized):
    _fields_ = ([("struct_size", u32)] + [(n, i32) for n in ("batch", "tokens", "channels", "grid", "kind", "y_split3")]
                + [(n, i64) for n in ("x_batch_stride", "x_token_stride", "res_batch_stride", "res_token_stride",
                                      "y_batch_stride", "y_token_stride", "mod_batch_stride", "w_batch_stride",
                                      "w_token_stride", "red_batch_stride")]
                + [(n, vp) for n in ("x_ptr", "in_index_ptr", "out_index_ptr", "gate_ptr", "scale_ptr", "shift_ptr",
                                     "residual_ptr", "y_ptr", "w_ptr", "wdot_ptr", "wsum_ptr", "tsum_ptr", "y_inv_scale_ptr")]
                + [("y_f16s_lds_offset", i32), ("reserved", i32)])


class XattnParams(_Sized):
    _fields_ = ([("struct_size", u32)] + [(n, i32) for n in ("batch", "seqlen", "heads", "head_dim")] + [("scale", f32), ("n_dirs", i32)]
                + [(n, i64) for n in ("qkv_batch_stride", "qkv_token_stride", "out_batch_stride", "out_token_stride")]
                + [(n, vp) for n in ("qkv1_ptr", "qkv2_ptr", "out_ptr", "lse_ptr", "bias1_ptr", "bias2_ptr")]
                + [("precision", i32), ("out_split3", i32)]
                + [(n, vp) for n in ("x1_inv_ptr", "x2_inv_ptr", "kv_bound_ptr", "out_inv_ptr")] + [("qkv_f16", i32), ("reserved2", i32)])


class XattnBwdParams(_Sized):
    _fields_ = ([("struct_size", u32), ("reserved", u32), ("fwd", XattnParams)] + [(n, i64) for n in ("dqkv_batch_stride", "dqkv_token_stride")]
                + [(n, vp) for n in ("dout_ptr", "dqkv1_ptr", "dqkv2_ptr", "delta_ptr")])


class GemmExt(_Sized):
    """dimsum_gemm_ext_t: fused-epilogue operands, operand-image read modes, timing, tuning"""
    _fields_ = ([("struct_size", u32), ("rows_per_batch", i32), ("timing_start_event", vp), ("timing_stop_event", vp)]
                + [(n, i32) for n in ("tune_variant", "tune_group_m", "tune_reserved", "c_image_pieces")]
                + [(n, vp) for n in ("gate_bound_ptr", "h_inv_scale_ptr", "residual_ptr", "gate_ptr")]
                + [("residual_ld", i64), ("gate_ld", i64), ("x12_ptr", vp), ("x12_ld", i64), ("a_alias_rows", i64), ("b_alias_rows", i64),
                   ("a_alias_weight_order", i32), ("qkv_q_cols", i32), ("conv_weight_ptr", vp), ("conv_bias_ptr", vp),
                   ("conv_rows", i32), ("conv_width", i32), ("conv_seq", i32), ("conv_weight_ld", i32),
                   ("a_block_inv_ptr", vp), ("a_block_inv_ld", i64), ("tn_pair_a_cols", i64), ("tn_pair_b_cols", i64),
                   ("k_scale_ptr", vp), ("c_scale_ptr", vp), ("k_inv_a_ptr", vp), ("k_inv_b_ptr", vp)])


class GemmParams(_Sized):
    _fields_ = ([("struct_size", u32)] + [(n, i32) for n in ("m", "n", "k", "operand_dtype", "epilogue")] + [("out_scale", f32), ("reserved", i32)]
                + [(n, i64) for n in ("lda", "ldb", "ldc")]
                + [(n, vp) for n in ("a_ptr", "b_ptr", "bias_ptr", "c_ptr", "a_inv_scale_ptr", "b_inv_scale_ptr")]
                + [("ext", C.POINTER(GemmExt))])


class F16sJob(C.Structure):
    _fields_ = ([(n, vp) for n in ("src", "dst", "inv_scale_ptr", "l1max_ptr", "absmax_ptr")]
                + [(n, i64) for n in ("rows", "cols", "src_row_stride", "dst_row_stride")] + [("l1_factor", f32), ("reserved", i32)])


GEMM_EPI_F32, GEMM_EPI_GATED_GELU_SPLIT3, GEMM_EPI_GATED_GELU_F16, GEMM_EPI_F32_BIAS, GEMM_EPI_F32_GATE_RESIDUAL, GEMM_EPI_F16_QKV, GEMM_EPI_F32_CONV = 0, 1, 2, 3, 4, 5, 6

# every symbol include/dimsum_hip.h declares (tests check the library exports all of them)
EXPORTS = (
    "dimsum_status_string", "dimsum_abi_version", "dimsum_target_arch",
    "dimsum_event_create", "dimsum_event_destroy", "dimsum_event_elapsed_ms",
    "dimsum_ssm_scan_fwd", "dimsum_ssm_scan_bwd", "dimsum_ssm_scan_bwd_workspace_bytes", "dimsum_ssm_scan_fwd_variant",
    "dimsum_causal_conv1d_fwd", "dimsum_causal_conv1d_bwd",
    "dimsum_norm_fwd", "dimsum_norm_bwd", "dimsum_token_transform", "dimsum_xattn_fusion_fwd", "dimsum_xattn_fusion_bwd",
    "dimsum_gated_gelu_fwd", "dimsum_gated_gelu_bwd", "dimsum_gated_gelu_fwd_split3", "dimsum_gated_gelu_bwd_split3", "dimsum_gated_gelu_bwd_pair", "dimsum_gated_gelu_bwd_f16s", "dimsum_split3", "dimsum_split3_t",
    "dimsum_gemm_nt", "dimsum_gemm_nt_kernel_for", "dimsum_gemm_tn", "dimsum_gemm_nn", "dimsum_row_factors", "dimsum_rows_block_f16s", "dimsum_rows_f16s", "dimsum_rows_f16s_multi",
)

_lib = None


def load():
    """Loads the HIP library. Raises RuntimeError (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"dimsum_amd: {LIB_PATH} not found. Build it with `python -c 'import __graft_entry__ as g; "
                           f"g.build()'` or `make -C dimsum_amd/csrc`. There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.dimsum_status_string.restype = C.c_char_p
    lib.dimsum_status_string.argtypes = [C.c_int]
    lib.dimsum_target_arch.restype = C.c_char_p
    lib.dimsum_abi_version.restype = C.c_int
    if hasattr(lib, "dimsum_event_create"):
        lib.dimsum_event_create.restype, lib.dimsum_event_create.argtypes = vp, []
        lib.dimsum_event_destroy.restype, lib.dimsum_event_destroy.argtypes = None, [vp]
        lib.dimsum_event_elapsed_ms.restype, lib.dimsum_event_elapsed_ms.argtypes = C.c_float, [vp, vp]
    for name, ptype in (("dimsum_ssm_scan_fwd", SsmParams), ("dimsum_ssm_scan_bwd", SsmBwdParams),
                        ("dimsum_causal_conv1d_fwd", ConvParams), ("dimsum_causal_conv1d_bwd", ConvBwdParams),
                        ("dimsum_norm_fwd", NormParams), ("dimsum_norm_bwd", NormBwdParams),
                        ("dimsum_token_transform", TtParams), ("dimsum_xattn_fusion_fwd", XattnParams),
                        ("dimsum_xattn_fusion_bwd", XattnBwdParams), ("dimsum_gemm_nt", GemmParams)):
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype = C.c_int
            fn.argtypes = [C.POINTER(ptype), vp]
    for name, nptr in (("dimsum_gated_gelu_fwd", 3), ("dimsum_gated_gelu_bwd", 5), ("dimsum_gated_gelu_fwd_split3", 3),
                       ("dimsum_gated_gelu_bwd_split3", 5), ("dimsum_gated_gelu_bwd_pair", 5)):
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype = C.c_int
            fn.argtypes = [vp] * nptr + [i64, i64, vp]
    if hasattr(lib, "dimsum_gated_gelu_bwd_f16s"):
        lib.dimsum_gated_gelu_bwd_f16s.restype = C.c_int
        lib.dimsum_gated_gelu_bwd_f16s.argtypes = [vp] * 6 + [i64, i64, vp]
    if hasattr(lib, "dimsum_gemm_nt_kernel_for"):
        lib.dimsum_gemm_nt_kernel_for.restype = C.c_int
        lib.dimsum_gemm_nt_kernel_for.argtypes = [C.POINTER(GemmParams)]
    if hasattr(lib, "dimsum_gemm_tn"):
        lib.dimsum_gemm_tn.restype = C.c_int
        lib.dimsum_gemm_tn.argtypes = [C.POINTER(GemmParams), i32, i64, vp]
    if hasattr(lib, "dimsum_rows_block_f16s"):
        lib.dimsum_rows_block_f16s.restype = C.c_int
        lib.dimsum_rows_block_f16s.argtypes = [vp, i64, i64, i64, vp, i64, vp, i64, vp]
    if hasattr(lib, "dimsum_row_factors"):
        lib.dimsum_row_factors.restype = C.c_int
        lib.dimsum_row_factors.argtypes = [vp, vp, i64, vp, vp, vp]
    if hasattr(lib, "dimsum_gemm_nn"):
        lib.dimsum_gemm_nn.restype = C.c_int
        lib.dimsum_gemm_nn.argtypes = [C.POINTER(GemmParams), i32, i64, vp]
    if hasattr(lib, "dimsum_split3"):
        lib.dimsum_split3.restype = C.c_int
        lib.dimsum_split3.argtypes = [vp, i64, i64, i64, vp, i32, vp]
    if hasattr(lib, "dimsum_split3_t"):
        lib.dimsum_split3_t.restype = C.c_int
        lib.dimsum_split3_t.argtypes = [vp, i64, i64, i64, vp, vp]
    if hasattr(lib, "dimsum_rows_f16s"):
        lib.dimsum_rows_f16s.restype = C.c_int
        lib.dimsum_rows_f16s.argtypes = [vp, i64, i64, i64, vp, i64, vp, vp, vp]
    if hasattr(lib, "dimsum_rows_f16s_multi"):
        lib.dimsum_rows_f16s_multi.restype = C.c_int
        lib.dimsum_rows_f16s_multi.argtypes = [C.POINTER(F16sJob), i32, vp]
    if hasattr(lib, "dimsum_ssm_scan_bwd_workspace_bytes"):
        lib.dimsum_ssm_scan_bwd_workspace_bytes.restype = i64
        lib.dimsum_ssm_scan_bwd_workspace_bytes.argtypes = [i32] * 5
    if hasattr(lib, "dimsum_ssm_scan_fwd_variant"):
        lib.dimsum_ssm_scan_fwd_variant.restype = C.c_int
        lib.dimsum_ssm_scan_fwd_variant.argtypes = [C.POINTER(SsmParams)]
    if lib.dimsum_abi_version() != 17:
        raise RuntimeError("dimsum_amd: libdimsum_hip.so ABI version mismatch; rebuild")
    _lib = lib
    return lib


GEMM_NT_KERNELS = {0: "gemm_nt_kernel<256 x 256 tiles>", 1: "gemm_nt_m128_kernel<128 x 256 tiles>", 2: "gemm_nt_persist_kernel"}   # dimsum_gemm_nt_kernel_for()
SCAN_FWD_KERNELS = {1: "ssm_scan_fwd_kernel", 2: "ssm_scan_fwd_split_kernel<2 lanes per channel>",
                    4: "ssm_scan_fwd_split_kernel<4 lanes per channel>",
                    16: "ssm_scan_fwd_lanes_kernel<one lane per state>"}      # dimsum_ssm_scan_fwd_variant() -> kernel


def check(status, what):
    if status != 0:
        raise RuntimeError(f"{what}: {load().dimsum_status_string(status).decode()} (status {status})")
