"""ctypes binding of libdiffews_hip.so (C ABI declared in include/diffews_hip.h).

There is NO fallback: if the library is missing this raises, and every op in `ops.py` goes
through it.  The product path never touches a CPU or plain-PyTorch implementation.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DFW_LIB") or os.path.join(_HERE, "libdiffews_hip.so")   # DFW_LIB: A/B another build

BF16, F16 = 0, 1
OUT_T, OUT_F32, OUT_NCHW_F32 = 0, 1, 2
ACT_NONE, ACT_SILU, ACT_CLAMP1 = 0, 1, 2

_vp, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t


class GemmArgs(C.Structure):
    _fields_ = [("A", _vp), ("W", _vp), ("C", _vp), ("bias", _vp), ("rowbias", _vp), ("residual", _vp),
                ("workspace", _vp), ("workspace_bytes", _sz), ("a_elems", _i64), ("w_elems", _i64),
                ("M", _i32), ("N", _i32), ("K", _i32), ("lda", _i32), ("ldc", _i32), ("ldr", _i32), ("ld_rowbias", _i32),
                ("taps", _i32), ("Cin", _i32), ("Hi", _i32), ("Wi", _i32), ("Ho", _i32), ("Wo", _i32),
                ("stride", _i32), ("pad", _i32), ("ups", _i32), ("rows_per_img", _i32),
                ("out_scale", _f32), ("act", _i32), ("geglu", _i32), ("out_mode", _i32), ("splitk", _i32),
                ("batch", _i32), ("strideA", _i64), ("strideW", _i64), ("strideC", _i64), ("dtype", _i32),
                ("gn_partial", _vp), ("gn_groups", _i32),
                ("colscale", _f32), ("colscale_n", _i32), ("residual_f32", _i32), ("W_blocked", _vp)]


class FsaArgs(C.Structure):
    _fields_ = [("q", _vp), ("k", _vp), ("v", _vp), ("k_bank", _vp), ("v_bank", _vp), ("out", _vp),
                ("batch", _i32), ("heads", _i32), ("n_q", _i32), ("n_kv", _i32), ("n_bank", _i32), ("nshot", _i32),
                ("ldq", _i32), ("ldk", _i32), ("ldv", _i32), ("ldkb", _i32), ("ldvb", _i32), ("ldo", _i32),
                ("q_bs", _i64), ("k_bs", _i64), ("v_bs", _i64), ("kb_bs", _i64), ("vb_bs", _i64), ("o_bs", _i64),
                ("scale", _f32), ("dtype", _i32), ("n_plain", _i32), ("q_prescaled", _i32), ("lse", _vp),
                ("workspace", _vp), ("workspace_bytes", _sz)]


class XattnArgs(C.Structure):
    _fields_ = [("q", _vp), ("k", _vp), ("v", _vp), ("out", _vp),
                ("batch", _i32), ("heads", _i32), ("n_q", _i32), ("L", _i32),
                ("ldq", _i32), ("ldk", _i32), ("ldv", _i32), ("ldo", _i32),
                ("q_bs", _i64), ("k_bs", _i64), ("v_bs", _i64), ("o_bs", _i64),
                ("scale", _f32), ("dtype", _i32)]


class GroupNormArgs(C.Structure):
    _fields_ = [("x", _vp), ("y", _vp), ("gamma", _vp), ("beta", _vp), ("stats_ws", _vp), ("stats_ws_bytes", _sz),
                ("B", _i32), ("HW", _i32), ("C", _i32), ("groups", _i32), ("ldx", _i32), ("ldy", _i32),
                ("eps", _f32), ("silu", _i32), ("dtype", _i32), ("pre_partial", _vp), ("pre_chunks", _i32),
                ("x_f32", _i32)]


class LayerNormArgs(C.Structure):
    _fields_ = [("x", _vp), ("y", _vp), ("gamma", _vp), ("beta", _vp),
                ("rows", _i32), ("C", _i32), ("ldx", _i32), ("ldy", _i32), ("eps", _f32), ("dtype", _i32),
                ("x_f32", _i32)]


class ConvSmallArgs(C.Structure):
    _fields_ = [("x", _vp), ("W", _vp), ("bias", _vp), ("y", _vp),
                ("B", _i32), ("Cin", _i32), ("H", _i32), ("Wd", _i32), ("Cout", _i32), ("taps", _i32), ("ldy", _i32),
                ("in_scale", _f32), ("out_scale", _f32), ("out_mode", _i32), ("dtype", _i32),
                ("gn_partial", _vp), ("gn_groups", _i32), ("y_bstride", _i64),
                ("x1", _vp), ("x2", _vp), ("b0", _i32), ("b1", _i32)]


class ImageArgs(C.Structure):
    _fields_ = [("src", _vp), ("H", _i32), ("W", _i32), ("out_h", _i32), ("out_w", _i32),
                ("xbounds", _vp), ("xcoef", _vp), ("xk", _i32), ("ybounds", _vp), ("ycoef", _vp), ("yk", _i32),
                ("tmp", _vp), ("dst", _vp), ("lut", _vp)]


class GemmTnArgs(C.Structure):
    _fields_ = [("A", _vp), ("B", _vp), ("out", _vp), ("workspace", _vp), ("workspace_bytes", _sz),
                ("a_elems", _i64), ("b_elems", _i64),
                ("M", _i32), ("N", _i32), ("Kc", _i32), ("lda", _i32), ("ldb", _i32),
                ("taps", _i32), ("Hi", _i32), ("Wi", _i32), ("Ho", _i32), ("Wo", _i32), ("stride", _i32), ("pad", _i32), ("ups", _i32),
                ("batch", _i32), ("batch2", _i32), ("strideA", _i64), ("strideB", _i64), ("strideA2", _i64), ("strideB2", _i64),
                ("ldo_n", _i64), ("ldo_t", _i64), ("ldo_b", _i64), ("scale", _f32), ("accumulate", _i32), ("dtype", _i32)]


class GroupNormBwdArgs(C.Structure):
    _fields_ = [("x", _vp), ("dy", _vp), ("dx", _vp), ("gamma", _vp), ("beta", _vp), ("mean_rstd", _vp),
                ("dgamma", _vp), ("dbeta", _vp), ("workspace", _vp), ("workspace_bytes", _sz),
                ("B", _i32), ("HW", _i32), ("C", _i32), ("groups", _i32), ("ldx", _i32), ("lddy", _i32), ("lddx", _i32),
                ("silu", _i32), ("accumulate", _i32), ("grad_scale", _f32), ("dtype", _i32), ("dx_add", _vp)]


class LayerNormBwdArgs(C.Structure):
    _fields_ = [("x", _vp), ("dy", _vp), ("dx", _vp), ("gamma", _vp), ("dgamma", _vp), ("dbeta", _vp),
                ("workspace", _vp), ("workspace_bytes", _sz),
                ("rows", _i32), ("C", _i32), ("ldx", _i32), ("lddy", _i32), ("lddx", _i32), ("eps", _f32),
                ("accumulate", _i32), ("grad_scale", _f32), ("dtype", _i32), ("dx_add", _vp)]


class FsaBwdArgs(C.Structure):
    _fields_ = [("qkv", _vp), ("out", _vp), ("dout", _vp), ("lse", _vp), ("delta", _vp), ("dqkv", _vp),
                ("batch", _i32), ("heads", _i32), ("n", _i32), ("nshot", _i32), ("n_plain", _i32),
                ("ld", _i32), ("ldo", _i32), ("ldd", _i32), ("scale", _f32), ("dtype", _i32),
                ("workspace", _vp), ("workspace_bytes", _sz), ("delta_bytes", _sz)]


class XattnBwdArgs(C.Structure):
    _fields_ = [("q", _vp), ("k", _vp), ("v", _vp), ("dout", _vp), ("dq", _vp), ("dk", _vp), ("dv", _vp),
                ("workspace", _vp), ("workspace_bytes", _sz),
                ("batch", _i32), ("heads", _i32), ("n_q", _i32), ("L", _i32),
                ("ldq", _i32), ("ldk", _i32), ("ldv", _i32), ("ldo", _i32), ("lddq", _i32), ("lddkv", _i32),
                ("q_bs", _i64), ("k_bs", _i64), ("v_bs", _i64), ("o_bs", _i64), ("dq_bs", _i64), ("dkv_bs", _i64),
                ("scale", _f32), ("dtype", _i32)]


class AttnBwdArgs(C.Structure):
    _fields_ = [("q", _vp), ("k", _vp), ("v", _vp), ("out", _vp), ("dout", _vp), ("lse", _vp), ("delta", _vp),
                ("dq", _vp), ("dk", _vp), ("dv", _vp),
                ("batch", _i32), ("heads", _i32), ("n_q", _i32), ("n_kv", _i32),
                ("ldq", _i32), ("ldkv", _i32), ("ldo", _i32), ("lddq", _i32), ("lddkv", _i32),
                ("q_bs", _i64), ("kv_bs", _i64), ("o_bs", _i64), ("dq_bs", _i64), ("dkv_bs", _i64),
                ("scale", _f32), ("dtype", _i32), ("workspace", _vp), ("workspace_bytes", _sz), ("delta_bytes", _sz)]


class VattnArgs(C.Structure):
    _fields_ = [("q", _vp), ("k", _vp), ("v", _vp), ("out", _vp),
                ("batch", _i32), ("n", _i32), ("head_dim", _i32), ("q_prescaled", _i32),
                ("ldq", _i32), ("ldk", _i32), ("ldv", _i32), ("ldo", _i32),
                ("q_bs", _i64), ("k_bs", _i64), ("v_bs", _i64), ("o_bs", _i64), ("dtype", _i32)]


class AdamWArgs(C.Structure):
    _fields_ = [("param", _vp), ("grad", _vp), ("exp_avg", _vp), ("exp_avg_sq", _vp), ("grad_sumsq", _vp), ("n", _i64),
                ("lr", _f32), ("beta1", _f32), ("beta2", _f32), ("eps", _f32), ("weight_decay", _f32), ("max_grad_norm", _f32),
                ("step", _i32), ("shadow", _vp), ("shadow_dtype", _i32), ("found_inf", _vp)]


class Config(C.Structure):
    _fields_ = [("conv_patch", _i32), ("big_kernels", _i32), ("big_bm", _i32), ("big_bn", _i32), ("big_bk", _i32),
                ("gemm_bm", _i32), ("gemm_bn", _i32), ("fsa_key_split", _i32),
                ("fsa_force_splits", _i32), ("big_min_tiles", _i32), ("k8", _i32)]


# every symbol include/diffews_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "dfw_configure": (_i32, [C.POINTER(Config)]),
    "dfw_get_config": (None, [C.POINTER(Config)]),
    "dfw_version": (_i32, []),
    "dfw_error_string": (C.c_char_p, [_i32]),
    "dfw_graph_memset_nodes": (_i32, [_vp, C.POINTER(_i32)]),
    "dfw_gemm": (_i32, [C.POINTER(GemmArgs), _vp]),
    "dfw_gemm_workspace_bytes": (_sz, [C.POINTER(GemmArgs)]),
    "dfw_gemm_kernel_name": (_i32, [C.POINTER(GemmArgs), C.c_char_p, _sz]),
    "dfw_gemm_gn_chunks": (_i32, [C.POINTER(GemmArgs)]),
    "dfw_fsa_attention": (_i32, [C.POINTER(FsaArgs), _vp]),
    "dfw_fsa_workspace_bytes": (_sz, [C.POINTER(FsaArgs)]),
    "dfw_cross_attention": (_i32, [C.POINTER(XattnArgs), _vp]),
    "dfw_vae_attention": (_i32, [C.POINTER(VattnArgs), _vp]),
    "dfw_groupnorm": (_i32, [C.POINTER(GroupNormArgs), _vp]),
    "dfw_groupnorm_workspace_bytes": (_sz, [C.POINTER(GroupNormArgs)]),
    "dfw_layernorm": (_i32, [C.POINTER(LayerNormArgs), _vp]),
    "dfw_conv_small": (_i32, [C.POINTER(ConvSmallArgs), _vp]),
    "dfw_conv_small_gn_chunks": (_i32, [C.POINTER(ConvSmallArgs)]),
    "dfw_softmax_rows": (_i32, [_vp, _vp, _i64, _i32, _f32, _i32, _vp]),
    "dfw_softmax_groups": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "dfw_transpose": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "dfw_concat_channels": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "dfw_convert_f32": (_i32, [_vp, _vp, _i64, _i32, _vp]),
    "dfw_zero": (_i32, [_vp, _i64, _vp]),
    "dfw_split_f32": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "dfw_timestep_embedding": (_i32, [_vp, _vp, _i32, _i32, _i32, _f32, _i32, _vp]),
    "dfw_seg_postprocess": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "dfw_seg_postprocess_ex": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _f32, _i32, _vp]),
    "dfw_meter_update": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "dfw_gemm_tn": (_i32, [C.POINTER(GemmTnArgs), _vp]),
    "dfw_gemm_tn_workspace_bytes": (_sz, [C.POINTER(GemmTnArgs)]),
    "dfw_colsum": (_i32, [_vp, _vp, _vp, _sz, _i64, _i32, _i32, _i32, _i64, _f32, _i32, _i32, _vp]),
    "dfw_colsum_plan": (_i32, [_i64, C.POINTER(_i32), C.POINTER(_i32)]),
    "dfw_table_write": (_i32, [_vp, _i64, _vp, _i32, _vp]),
    "dfw_colsum_batch": (_i32, [_vp, _i32, _i64, _i64, _vp, _i32, _vp]),
    "dfw_colsum_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "dfw_groupnorm_bwd": (_i32, [C.POINTER(GroupNormBwdArgs), _vp]),
    "dfw_groupnorm_bwd_workspace_bytes": (_sz, [C.POINTER(GroupNormBwdArgs)]),
    "dfw_layernorm_bwd": (_i32, [C.POINTER(LayerNormBwdArgs), _vp]),
    "dfw_layernorm_bwd_workspace_bytes": (_sz, [_i32, _i32]),
    "dfw_geglu": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "dfw_elementwise": (_i32, [_i32, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dfw_nchw_to_nhwc": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _f32, _i32, _vp]),
    "dfw_mse_loss": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _i32, _vp]),
    "dfw_loss_grad": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _f32, _i32, _vp]),
    "dfw_convert_to_f32": (_i32, [_vp, _vp, _i64, _f32, _i32, _vp]),
    "dfw_fsa_attention_bwd": (_i32, [C.POINTER(FsaBwdArgs), _vp]),
    "dfw_fsa_attention_bwd_workspace_bytes": (_sz, [C.POINTER(FsaBwdArgs)]),
    "dfw_attention_bwd": (_i32, [C.POINTER(AttnBwdArgs), _vp]),
    "dfw_attention_bwd_workspace_bytes": (_sz, [C.POINTER(AttnBwdArgs)]),
    "dfw_cross_attention_bwd": (_i32, [C.POINTER(XattnBwdArgs), _vp]),
    "dfw_cross_attention_bwd_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "dfw_silu": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "dfw_sumsq": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "dfw_adamw": (_i32, [C.POINTER(AdamWArgs), _vp]),
    "dfw_weight_relayout": (_i32, [_vp, _vp, _i32, _i32, _i64, _i64, _i32, _i64, _i64, _i32, _vp]),
    "dfw_weight_relayout_batch": (_i32, [_vp, _i32, _i64, _vp]),
    "dfw_resample_ksize": (_i32, [_i32, _i32]),
    "dfw_resample_coeffs": (_i32, [_i32, _i32, _vp, _vp]),
    "dfw_image_to_tensor": (_i32, [C.POINTER(ImageArgs), _vp]),
    "dfw_mask_to_tensor": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
}

_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library is not built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the MI355X kernels are not built. Run `python -m diffews_amd.build` "
                "(or __graft_entry__.build()). There is no CPU / PyTorch fallback.")
        # torch must be loaded first: it ships its own libamdhip64 and the kernels here run on
        # torch's streams, so both have to bind to the SAME HIP runtime instance (loading this
        # library first pulls /opt/rocm's copy in and the process ends up with no visible device).
        import torch  # noqa: F401
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def configure(**fields):
    """Change fields of the library's process-wide tuning record (dfw_config): sweeps and A/B runs only -- the defaults
    are the measured-best plans.  configure() with no arguments restores the defaults.  Returns the record in effect."""
    h = lib()
    if not fields:
        check(h.dfw_configure(None), "dfw_configure")
    cfg = Config()
    h.dfw_get_config(C.byref(cfg))
    for k, v in fields.items():
        if not hasattr(cfg, k):
            raise AttributeError(f"dfw_config has no field {k!r}")
        setattr(cfg, k, int(v))
    if fields:
        check(h.dfw_configure(C.byref(cfg)), "dfw_configure")
    return {k: getattr(cfg, k) for k, _ in Config._fields_}


def check(rc, what):
    if rc != 0:
        msg = lib().dfw_error_string(rc)
        raise RuntimeError(f"{what} failed: [{rc}] {msg.decode() if msg else '?'}")
