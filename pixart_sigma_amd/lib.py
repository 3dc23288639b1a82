"""ctypes binding of libpixart_hip.so (include/pixart_hip.h).  Python-side mirror of the C ABI: tensors are passed as
raw device pointers + sizes + the current HIP stream; nothing here computes.  The product path FAILS LOUDLY if the
library is missing or a call returns an error — there is no CPU / eager-PyTorch fallback."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# 16-bit operand type of the process: bf16 (default; training) or IEEE fp16 (PXA_OPERAND_DTYPE=f16: the reference's inference dtype,
# the build that meets the 1e-3 forward-parity tolerance).  One library per type, built from the same sources.
OPERAND = os.environ.get("PXA_OPERAND_DTYPE", "bf16").lower()
assert OPERAND in ("bf16", "f16"), f"PXA_OPERAND_DTYPE must be bf16 or f16, got {OPERAND!r}"
OPERAND_DTYPE = torch.float16 if OPERAND == "f16" else torch.bfloat16
LIB_PATH = os.environ.get("PXA_LIB_PATH") or os.path.join(_HERE, "libpixart_hip_f16.so" if OPERAND == "f16" else "libpixart_hip.so")   # env override: A/B kernel builds
ABI_VERSION = 9

c_void_p, c_int, c_long, c_float = C.c_void_p, C.c_int, C.c_long, C.c_float


class GemmArgs(C.Structure):
    _fields_ = [("A", c_void_p), ("B", c_void_p), ("lda", c_int), ("ldb", c_int),
                ("M", c_int), ("N", c_int), ("K", c_int), ("layout", c_int),
                ("bias", c_void_p), ("act", c_int), ("aux", c_void_p), ("ldaux", c_int),
                ("out_bf16", c_void_p), ("out2_bf16", c_void_p), ("ld_out", c_int),
                ("out_f32", c_void_p), ("ld_f32", c_int), ("accumulate", c_int), ("split_k", c_int),
                ("splitk_ws", c_void_p), ("splitk_ws_elems", c_long), ("colsum", c_void_p), ("colsum_stride", c_long),
                ("k_seg", c_int), ("a_seg_stride", c_long), ("k_tap", c_int),
                ("gn_part", c_void_p), ("gn_img_rows", c_int), ("gn_row_pitch", c_int), ("gn_h", c_int), ("gn_w", c_int), ("items_descending", c_int),
                ("up_row_pitch", c_int), ("up_img_rows", c_int), ("up_dy", c_int), ("up_dx", c_int)]


class GridArg(C.Structure):
    """pxa_grid: a bf16 NHWC pixel grid (include/pixart_hip.h, VAE conv stack)."""
    _fields_ = [("ptr", c_void_p), ("B", c_int), ("H", c_int), ("W", c_int), ("C", c_int),
                ("row_pitch", c_int), ("img_pitch", c_long), ("origin", c_long)]


class CameTensor(C.Structure):
    _fields_ = [("off", c_long), ("batch", c_int), ("R", c_int), ("C", c_int), ("factored", c_int),
                ("row_off", c_long), ("col_off", c_long), ("rm_off", c_long), ("nf_off", c_long)]


class CameTile(C.Structure):
    _fields_ = [("tensor", c_int), ("first", c_int), ("count", c_int), ("pad", c_int)]


class CameArgs(C.Structure):
    _fields_ = [("p", c_void_p), ("g", c_void_p), ("exp_avg", c_void_p), ("p_bf16", c_void_p),
                ("sq_row", c_void_p), ("sq_col", c_void_p), ("res_row", c_void_p), ("res_col", c_void_p), ("nf_sq", c_void_p),
                ("scratch", c_void_p), ("tensors", c_void_p), ("n_tensors", c_int), ("tiles", c_void_p), ("n_tiles", c_int),
                ("col_inv_r", c_void_p), ("n_cols_total", c_long), ("n_rm_total", c_long),
                ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("beta3", C.c_double), ("eps0", C.c_double), ("eps1", C.c_double),
                ("clip_threshold", C.c_double), ("weight_decay", C.c_double), ("gscale", c_void_p), ("scaler", c_void_p)]


class AttnArgs(C.Structure):
    _fields_ = [("q", c_void_p), ("k", c_void_p), ("v", c_void_p), ("o", c_void_p),
                ("d_o", c_void_p), ("dq", c_void_p), ("dk", c_void_p), ("dv", c_void_p),
                ("lse", c_void_p), ("delta", c_void_p),
                ("q_bs", c_long), ("q_ts", c_long), ("k_bs", c_long), ("k_ts", c_long),
                ("v_bs", c_long), ("v_ts", c_long), ("o_bs", c_long), ("o_ts", c_long),
                ("q_hs", c_int), ("k_hs", c_int), ("v_hs", c_int), ("o_hs", c_int),
                ("dq_bs", c_long), ("dq_ts", c_long), ("dk_bs", c_long), ("dk_ts", c_long), ("dv_bs", c_long), ("dv_ts", c_long),
                ("dq_hs", c_int), ("dk_hs", c_int), ("dv_hs", c_int),
                ("B", c_int), ("H", c_int), ("Nq", c_int), ("Nk", c_int), ("head_dim", c_int),
                ("kv_start", c_void_p), ("kv_len", c_void_p), ("max_kv_len", c_int), ("scale", c_float),
                ("dq_colsum", c_void_p), ("dk_colsum", c_void_p), ("dv_colsum", c_void_p), ("colsum_stride", c_long),
                ("bwd_stats", c_void_p), ("q_prescaled", c_int)]


# name -> argtypes (all return int); must list every symbol include/pixart_hip.h declares
_P, _I, _L, _F = c_void_p, c_int, c_long, c_float
_G = C.POINTER(GridArg)
SIGNATURES = {
    "pxa_gemm": [C.POINTER(GemmArgs), _P],
    "pxa_ln_mod_fwd": [_P, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P],
    "pxa_ln_mod_bwd": [_P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _P, _L, _I, _I, _I, _P],
    "pxa_ln_affine_fwd": [_P, _L, _P, _P, _P, _L, _P, _P, _P, _I, _I, _F, _P],
    "pxa_ln_affine_bwd": [_P, _L, _P, _P, _P, _P, _P, _L, _P, _P, _I, _I, _P],
    "pxa_gate_bwd": [_P, _P, _P, _P, _I, _P, _P, _P, _I, _P, _L, _I, _I, _I, _P],
    "pxa_colsum_reduce": [_P, _L, _P, _L, _P],
    "pxa_colsum_bf16": [_P, _I, _P, _I, _I, _P],
    "pxa_attn_fwd": [C.POINTER(AttnArgs), _P],
    "pxa_attn_bwd": [C.POINTER(AttnArgs), _P],
    "pxa_patch_embed_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "pxa_patch_embed_bwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "pxa_unpatchify_fwd": [_P, _P, _I, _I, _I, _I, _P],
    "pxa_patchify_bwd": [_P, _P, _I, _I, _I, _I, _P],
    "pxa_gather_rows_bf16": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "pxa_kv_compress_fwd": [_P, _L, _L, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    "pxa_kv_compress_bwd": [_P, _P, _L, _L, _P, _P, _P, _P, _L, _L, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    "pxa_kv_pick": [_I, _P, _P, _L, _L, _I, _I, _I, _I, _I, _P],
    "pxa_sumsq_f32": [_P, _L, _P, _P],
    "pxa_clip_coef": [_P, _P, _F, _F, _P],
    "pxa_adamw_step": [_P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _P, _P],
    "pxa_cast_f32_bf16": [_P, _P, _L, _P],
    "pxa_scale_copy_f32": [_P, _L, _P, _P, _L, _I, _L, _L, _F, _P],
    "pxa_linear_f32_fwd": [_P, _P, _P, _P, _I, _I, _I, _P],
    "pxa_linear_f32_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "pxa_clip_coef_scaled": [_P, _P, _F, _F, _P, _F, _F, _I, _P],
    "pxa_adamw_step_scaled": [_P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _P, _P, _P],
    "pxa_came_step": [C.POINTER(CameArgs), _P],
    "pxa_iddpm_loss_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P],
    "pxa_iddpm_loss_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P],
    "pxa_vae_gn_stats": [_G, _I, _F, _P, _P, _P, _P],
    "pxa_vae_gn_finalize": [_P, _I, _I, _I, _L, _F, _P, _P, _P],
    "pxa_vae_gn_apply": [_G, _P, _P, _P, _P, _I, _I, _I, _G, _P],
    "pxa_vae_im2col3x3": [_G, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P],
    "pxa_vae_add": [_G, _G, _G, _P],
    "pxa_vae_softmax_rows": [_P, _L, _P, _L, _I, _I, _F, _P],
    "pxa_vae_nchw_to_grid": [_P, _I, _F, _G, _P],
    "pxa_vae_grid_to_nchw": [_G, _I, _P, _P],
    "pxa_vae_conv3x3_small_out": [_G, _P, _P, _P, _P, _I, _I, _P, _P, _I, _P, _P],
}
OTHER_SYMBOLS = ["pxa_last_error", "pxa_abi_version", "pxa_operand_dtype", "pxa_device_info", "pxa_gemm_splitk_ws_elems", "pxa_came_scratch_elems", "pxa_attn_bwd_stats_bytes", "pxa_gemm_set_dynamic_items",
                 "pxa_mfma_rate_probe_bytes", "pxa_mfma_rate_probe"]

_lib = None


class PixartHipError(RuntimeError):
    pass


def load():
    """Load the shared library once; raises if it has not been built (python -m pixart_sigma_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PixartHipError(f"{LIB_PATH} is missing: build it with `python -m pixart_sigma_amd.build` "
                             "(the PixArt-Sigma HIP path has no CPU/PyTorch fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = argtypes, c_int
    lib.pxa_last_error.restype = C.c_char_p
    lib.pxa_abi_version.restype = c_int
    lib.pxa_came_scratch_elems.argtypes, lib.pxa_came_scratch_elems.restype = [c_long, c_long, c_int], c_long
    lib.pxa_device_info.argtypes, lib.pxa_device_info.restype = [C.POINTER(c_int), C.POINTER(c_int)], c_int
    lib.pxa_attn_bwd_stats_bytes.argtypes, lib.pxa_attn_bwd_stats_bytes.restype = [c_int, c_int, c_int], c_long
    lib.pxa_gemm_set_dynamic_items.argtypes, lib.pxa_gemm_set_dynamic_items.restype = [c_int], c_int
    lib.pxa_mfma_rate_probe_bytes.argtypes, lib.pxa_mfma_rate_probe_bytes.restype = [], c_long
    lib.pxa_mfma_rate_probe.argtypes, lib.pxa_mfma_rate_probe.restype = [c_void_p, c_int, c_int, c_void_p, C.POINTER(C.c_double), c_void_p], c_int
    if lib.pxa_abi_version() != ABI_VERSION:
        raise PixartHipError(f"ABI mismatch: library {lib.pxa_abi_version()} vs binding {ABI_VERSION}")
    lib.pxa_operand_dtype.restype = c_int
    if lib.pxa_operand_dtype() != (1 if OPERAND == "f16" else 0):
        raise PixartHipError(f"{LIB_PATH} was built for the other operand type (PXA_OPERAND_DTYPE={OPERAND})")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise PixartHipError(f"{what} failed (rc={rc}): {load().pxa_last_error().decode()}")


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_cuda, "pixart_sigma_amd ops need tensors on the MI355X (no CPU fallback)"
    return c_void_p(t.data_ptr())


def call(name, *args):
    check(getattr(load(), name)(*args, stream()), name)
