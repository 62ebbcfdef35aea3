"""ctypes binding of ``libpanfusion_hip.so`` (C ABI in ``include/panfusion_hip.h``).

The product path has NO fallback: if the shared library is missing or a symbol
is absent, importing/using the ops raises.  Build it with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C panfusion_amd/csrc``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PF_HIP_LIB") or os.path.join(_HERE, "libpanfusion_hip.so")   # PF_HIP_LIB: A/B another build

PF_OK = 0
PF_BF16, PF_F16, PF_F32, PF_U8 = 0, 1, 2, 3

c_void_p, c_int, c_long, c_float, c_size_t = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_size_t
c_dp = C.POINTER(C.c_double)


class ConvDesc(C.Structure):
    """pf_conv_desc"""
    _fields_ = [
        ("a0", c_void_p), ("a1", c_void_p), ("c0", c_int), ("c1", c_int),
        ("a0_ld", c_int), ("a1_ld", c_int),
        ("n_img", c_int), ("h_in", c_int), ("w_in", c_int), ("h_out", c_int), ("w_out", c_int),
        ("ksize", c_int), ("stride", c_int), ("pad", c_int), ("upsample", c_int),
        ("w", c_void_p), ("n_out", c_int),
        ("bias", c_void_p), ("rowvec", c_void_p), ("rowvec_ld", c_int),
        ("residual", c_void_p), ("res_ld", c_int), ("res_dtype", c_int),
        ("out", c_void_p), ("out_ld", c_int), ("out_dtype", c_int), ("dtype", c_int),
        ("batch", c_int),
        ("a_bstride", c_long), ("w_bstride", c_long), ("out_bstride", c_long), ("res_bstride", c_long),
        ("epilogue", c_int), ("workspace", c_void_p), ("workspace_bytes", c_size_t),
        ("gn_partial", c_void_p), ("wrap_pad", c_int), ("crop", c_int),
        ("tickets", c_void_p), ("n_tickets", c_int), ("split3", c_int), ("subpixel", c_int),
    ]


class AttnDesc(C.Structure):
    """pf_attn_desc"""
    _fields_ = [
        ("q", c_void_p), ("k", c_void_p), ("vt", c_void_p), ("out", c_void_p),
        ("dtype", c_int), ("B", c_int), ("H", c_int), ("D", c_int), ("nq", c_int), ("nk", c_int),
        ("q_ld", c_int), ("k_ld", c_int), ("vt_ld", c_int), ("o_ld", c_int),
        ("q_bs", c_long), ("k_bs", c_long), ("vt_bs", c_long), ("o_bs", c_long),
        ("scale", c_float),
        ("bias", c_void_p), ("bias_ld", c_long), ("flags", c_void_p), ("flags_ld", c_int),
        ("lse", c_void_p),
        ("workspace", c_void_p), ("workspace_bytes", c_size_t),
    ]


class LinearWsDesc(C.Structure):
    """pf_linear_ws_desc"""
    _fields_ = [
        ("a", c_void_p), ("a_ld", c_int), ("w", c_void_p), ("bias", c_void_p),
        ("residual", c_void_p), ("res_ld", c_int), ("out", c_void_p), ("out_ld", c_int),
        ("out_vt", c_void_p), ("vt_ld", c_int), ("rows_per_batch", c_int), ("vt_bs", c_long),
        ("ln_gamma", c_void_p), ("ln_beta", c_void_p), ("ln_eps", c_float), ("ln_out", c_void_p), ("ln_ld", c_int),
        ("M", c_int), ("N", c_int), ("K", c_int), ("dtype", c_int), ("mode", c_int),
    ]


class AttnBwdDesc(C.Structure):
    """pf_attn_bwd_desc"""
    _fields_ = [
        ("q", c_void_p), ("k", c_void_p), ("v", c_void_p), ("dout", c_void_p),
        ("qt", c_void_p), ("kt", c_void_p), ("dot", c_void_p),
        ("dq", c_void_p), ("dk", c_void_p), ("dv", c_void_p),
        ("dtype", c_int), ("B", c_int), ("H", c_int), ("D", c_int), ("nq", c_int), ("nk", c_int),
        ("q_ld", c_int), ("k_ld", c_int), ("v_ld", c_int), ("do_ld", c_int),
        ("qt_ld", c_int), ("kt_ld", c_int), ("dot_ld", c_int),
        ("dq_ld", c_int), ("dk_ld", c_int), ("dv_ld", c_int),
        ("q_bs", c_long), ("k_bs", c_long), ("v_bs", c_long), ("do_bs", c_long), ("qt_bs", c_long), ("kt_bs", c_long),
        ("dot_bs", c_long), ("dq_bs", c_long), ("dk_bs", c_long), ("dv_bs", c_long),
        ("scale", c_float),
        ("bias", c_void_p), ("bias_ld", c_long), ("flags", c_void_p), ("flags_ld", c_int),
        ("lse", c_void_p), ("delta", c_void_p),
        ("workspace", c_void_p), ("workspace_bytes", c_size_t),
    ]


# name -> (restype, argtypes); must list every symbol include/panfusion_hip.h declares
SIGNATURES = {
    "pf_version": (c_int, []),
    "pf_last_error_string": (C.c_char_p, []),
    "pf_e2p_grid": (c_int, [c_dp, c_dp, c_dp, c_int, c_int, c_int, c_int, c_int,
                            c_void_p, c_void_p, c_void_p, c_void_p]),
    "pf_p2e_grid": (c_int, [c_dp, c_dp, c_dp, c_int, c_int, c_int, c_int, c_int,
                            c_void_p, c_void_p, c_void_p, c_void_p]),
    "pf_py360_e2p": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_dp, c_dp, c_dp, c_dp, c_dp, c_int, c_int, c_int, c_int,
                             c_void_p, c_void_p]),
    "pf_nearest_indices": (c_int, [c_void_p, c_void_p, c_long, c_int, c_int, c_void_p, c_void_p]),
    "pf_remap": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                         c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pf_equi_coords": (c_int, [c_int, c_int, c_void_p, c_void_p]),
    "pf_spherical_pe": (c_int, [c_void_p, c_long, c_void_p, c_int, c_void_p, c_void_p]),
    "pf_epa_tables_workspace_size": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "pf_epa_tables_build": (c_int, [c_dp, c_dp, c_dp, c_int, c_int, c_int, c_int, c_int,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pf_groupnorm_workspace_size": (c_size_t, [c_int, c_int, c_int]),
    "pf_groupnorm_stats": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pf_groupnorm_stats_wrap": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pf_scale_shift_act": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                   c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pf_layernorm": (c_int, [c_void_p, c_void_p, c_long, c_int, c_long, c_int, c_void_p, c_void_p,
                             c_float, c_int, c_void_p, c_void_p]),
    "pf_axpby": (c_int, [c_void_p, c_void_p, c_float, c_float, c_long, c_void_p, c_void_p]),
    "pf_vae_sample": (c_int, [c_void_p, c_void_p, c_int, c_int, c_long, c_float, c_void_p, c_void_p]),
    "pf_geglu": (c_int, [c_void_p, c_int, c_long, c_int, c_void_p, c_void_p]),
    "pf_timestep_features": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pf_timestep_features_strided": (c_int, [c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pf_silu": (c_int, [c_void_p, c_int, c_long, c_void_p, c_void_p]),
    "pf_pad_width": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pf_crop_width": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pf_pad_width_rows": (c_int, [c_void_p, c_int, c_long, c_int, c_int, c_void_p, c_void_p]),
    "pf_roll_width_rows": (c_int, [c_void_p, c_int, c_long, c_int, c_int, c_void_p, c_void_p]),
    "pf_nchw_to_nhwc": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pf_nhwc_to_nchw": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pf_add": (c_int, [c_void_p, c_int, c_void_p, c_int, c_long, c_void_p, c_void_p]),
    "pf_embed_tokens": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_long, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "pf_softmax_rows": (c_int, [c_void_p, c_long, c_int, c_long, c_float, c_int, c_void_p, c_long, c_void_p]),
    "pf_tensor_to_image": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pf_cfg_ddim_step": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_float,
                                 c_long, c_int, c_int, c_void_p, c_void_p]),
    "pf_cfg_ddim_step_pair": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_float,
                                      c_long, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, C.c_int64, c_void_p]),
    "pf_conv_gemm": (c_int, [C.POINTER(ConvDesc), c_void_p]),
    "pf_conv_gemm_workspace_size": (c_size_t, [C.POINTER(ConvDesc)]),
    "pf_conv_gemm_gn_rows": (c_int, [C.POINTER(ConvDesc)]),
    "pf_conv_gemm_kernel_id": (c_int, [C.POINTER(ConvDesc)]),
    "pf_groupnorm_from_partials": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pf_scale_shift_act_pair": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                                        c_void_p, c_void_p, c_void_p]),
    "pf_debug_gemm_profile": (c_int, [c_void_p, c_long]),
    "pf_linear_ws_supported": (c_int, [c_long, c_int, c_int, c_int]),
    "pf_linear_ws": (c_int, [C.POINTER(LinearWsDesc), c_void_p]),
    "pf_conv_in": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                           c_void_p, c_void_p]),
    "pf_conv_out": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                            c_void_p, c_void_p]),
    "pf_conv_out_gn": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                               c_void_p, c_void_p]),
    "pf_attention": (c_int, [C.POINTER(AttnDesc), c_void_p]),
    "pf_attention_workspace_size": (c_size_t, [C.POINTER(AttnDesc)]),
    "pf_attention_delta": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_long, c_int, c_long, c_void_p, c_void_p]),
    "pf_attention_bwd": (c_int, [C.POINTER(AttnBwdDesc), c_void_p]),
    "pf_attention_bwd_workspace_size": (c_size_t, [C.POINTER(AttnBwdDesc)]),
    "pf_layernorm_bwd_parts": (c_int, [c_long]),
    "pf_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_long, c_int, c_long, c_int, c_void_p, c_float, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p]),
    "pf_groupnorm_bwd_workspace_size": (c_size_t, [c_int, c_int, c_int]),
    "pf_groupnorm_bwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p,
                                 c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pf_groupnorm_param_grads_workspace_size": (c_size_t, [c_int, c_int, c_int]),
    "pf_groupnorm_param_grads": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pf_silu_bwd": (c_int, [c_void_p, c_int, c_void_p, c_long, c_void_p, c_void_p]),
    "pf_im2col3": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pf_lora_fold": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p, c_long, c_void_p, c_long,
                             c_void_p, c_long, c_void_p, c_long, c_void_p]),
    "pf_weighted_colsum_workspace_size": (c_size_t, [c_long, c_int, c_int]),
    "pf_weighted_colsum": (c_int, [c_void_p, c_int, c_long, c_int, c_long, c_void_p, c_int, c_long, c_void_p, c_float, c_void_p, c_int,
                                   c_void_p, c_void_p, c_size_t, c_void_p]),
    "pf_zero_insert2": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pf_sum2x2": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pf_pad_width_bwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pf_crop_width_bwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pf_transpose_tokens": (c_int, [c_void_p, c_int, c_int, c_long, c_int, c_void_p, c_void_p]),
    "pf_geglu_bwd": (c_int, [c_void_p, c_void_p, c_int, c_long, c_int, c_void_p, c_void_p]),
    "pf_colsum_workspace_size": (c_size_t, [c_long, c_int]),
    "pf_colsum": (c_int, [c_void_p, c_int, c_long, c_int, c_long, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pf_amax_f32": (c_int, [c_void_p, c_long, c_void_p, c_int, c_void_p]),
    "pf_pow2_scale": (c_int, [c_void_p, c_void_p]),
    "pf_scale_f32": (c_int, [c_void_p, c_long, c_void_p, c_int, c_int, c_void_p, c_void_p]),
}


class PanFusionHipError(RuntimeError):
    pass


_lib = None


def lib():
    """Load (once) and return the CDLL with typed entry points.  Raises if the
    extension has not been built -- there is deliberately no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PanFusionHipError(
            "HIP extension %s not built; run `make -C panfusion_amd/csrc` "
            "(or __graft_entry__.build()).  There is no CPU fallback." % LIB_PATH)
    # Load torch FIRST: the PyTorch-ROCm wheel ships its own HIP runtime (libamdhip64); the library
    # must resolve to that same runtime instance, or its launches see "no ROCm-capable device".
    import torch  # noqa: F401
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)      # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = handle
    return handle


def check(status, what=""):
    if status != PF_OK:
        msg = lib().pf_last_error_string()
        raise PanFusionHipError("%s failed (status %d): %s" % (what, status, msg.decode() if msg else ""))


def source_hash():
    """sha256 (first 16 hex digits) over the kernel sources the library is built from (csrc/*.hip, *.h, Makefile and the C-ABI header):
    committed rocprof summaries carry it (tools/prof_summary.py, tools/gpu_mfma_instep.sh) and bench.py refuses to quote one that was
    measured on other sources."""
    import glob
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "csrc", "*.hip")) + glob.glob(os.path.join(here, "csrc", "*.h")) +
                   [os.path.join(here, "csrc", "Makefile"), os.path.join(os.path.dirname(here), "include", "panfusion_hip.h")])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
