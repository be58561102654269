"""ctypes binding of libvlg_hip.so (C ABI declared in include/vlg_hip.h).

There is deliberately NO fallback: if the shared library is missing or a kernel
launch reports an error this module raises.  The product path never computes on
the CPU and never imports anything from oracle/.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# VLG_HIP_LIB: development override (A/B runs of two builds in one gpurun call, tools/kernel_bench.py); still no fallback
LIB_PATH = os.environ.get("VLG_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "libvlg_hip.so")

EPI_NONE, EPI_BIAS, EPI_GELU, EPI_RESID, EPI_DGELU, EPI_BF16 = 0, 1, 2, 4, 8, 16
EPI_A_BF16, EPI_B_BF16, EPI_OUT_BF16, EPI_SPLIT3 = 32, 64, 128, 256      # bf16 activation storage (include/vlg_hip.h)
EPI_ACT_GELU = 512                                                       # activation operand = gelu(stored), applied on load
EPI_GELU_GRAD, EPI_MUL = 1024, 2048                                      # aux_out = gelu'(pre);  dgrad: C = acc * aux_in

P, I, L, F = c_void_p, c_int, c_int64, c_float

# name -> (restype, argtypes); must list every symbol include/vlg_hip.h declares
SIGNATURES = {
    "vlg_abi_version": (I, []),
    "vlg_build_arch": (c_char_p, []),
    "vlg_embed_fwd": (I, [P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "vlg_embed_bwd_slabs": (I, []),
    "vlg_embed_bwd_slabs_for": (I, [I, I, I, I, I]),
    "vlg_embed_bwd": (I, [P, P, P, P, L, L, I, I, I, I, I, P]),
    "vlg_layernorm_fwd": (I, [P, P, P, P, P, P, L, I, F, P]),
    "vlg_layernorm_fwd_bf16": (I, [P, P, P, P, P, P, L, I, F, P]),
    "vlg_layernorm_bwd_slabs": (I, [L]),
    "vlg_layernorm_bwd": (I, [P, P, P, P, P, P, P, P, L, L, L, I, P]),
    "vlg_layernorm_bwd_bf16": (I, [P, P, P, P, P, P, P, P, L, L, L, I, P]),
    "vlg_linear_fwd": (I, [P, I, P, I, P, P, I, P, P, L, I, I, I, P]),
    "vlg_linear_dgrad": (I, [P, I, P, I, P, I, P, L, I, I, I, P]),
    "vlg_linear_wgrad_slabs": (I, [L, I, I]),
    "vlg_linear_dgrad_wgrad": (I, [P, I, P, I, P, I, P, P, I, P, L, L, L, I, I, I, P, I, P]),
    "vlg_linear_wgrad_slabs_for": (I, [L, I, I, I]),
    "vlg_linear_wgrad": (I, [P, I, P, I, P, L, L, L, I, I, I, P]),
    "vlg_attention_fwd": (I, [P, P, L, I, I, P]),
    "vlg_attention_bwd": (I, [P, P, P, L, I, I, P]),
    "vlg_attention_fwd_bf16": (I, [P, P, L, I, I, P]),
    "vlg_attention_bwd_bf16": (I, [P, P, P, L, I, I, P]),
    "vlg_attention_clip_fwd": (I, [P, P, P, P, L, I, I, I, P]),
    "vlg_attention_clip_bwd": (I, [P, P, P, P, P, P, P, L, I, I, I, P]),
    "vlg_layout_loss_scratch": (I, []),
    "vlg_layout_loss": (I, [P, I, P, P, P, P, P, P, I, I, I, I, F, F, F, F, F, P]),
    "vlg_reduce_slabs": (I, [P, L, I, P, L, P]),
    "vlg_reduce_slabs_table": (I, [P, I, I, P]),
    "vlg_sum_partials_table": (I, [P, I, P]),
    "vlg_adam_step": (I, [P, P, P, P, L, I, F, F, F, F, F, P]),
    "vlg_adam_step_graph": (I, [P, P, P, P, P, L, P, I, F, F, F, F, F, P]),
    "vlg_adam_step_bf16": (I, [P, P, P, P, P, L, I, F, F, F, F, F, P]),
    "vlg_image_loss_scratch": (I, []),
    "vlg_ce_nchw": (I, [P, P, P, P, P, I, I, L, F, P]),
    "vlg_l1_mean": (I, [P, P, P, P, P, L, F, P]),
    "vlg_gradient_loss": (I, [P, P, P, P, P, I, I, I, F, P]),
    "vlg_ssim_loss": (I, [P, P, P, P, P, I, I, I, I, F, P]),
    "vlg_affine_nchw": (I, [P, P, I, I, L, P, P, P]),
    "vlg_prep_input": (I, [P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, P]),
    "vlg_argmax_nchw": (I, [P, P, I, I, L, P]),
    "vlg_rollout_input": (I, [P, P, P, P, P, P, P, I, L, P]),
    "vlg_conv3x3_fwd": (I, [P, P, P, P, P, P, P, P, L, I, I, I, I, I, I, P, L, P]),
    "vlg_conv3x3_fwd_splits": (I, [L, I, I, I]),
    "vlg_conv3x3_fwd_workspace": (L, [L, I, I, I]),
    "vlg_conv3x3_dgrad_slabs": (I, [L, I]),
    "vlg_conv3x3_dgrad": (I, [P, P, P, P, P, P, P, P, L, L, I, I, I, I, I, P, L, I, P]),
    "vlg_conv3x3_dgrad_splits": (I, [L, I, I]),
    "vlg_conv3x3_dgrad_workspace": (L, [L, I, I]),
    "vlg_conv3x3_wgrad_slabs": (I, [L, I, I]),
    "vlg_conv3x3_wgrad": (I, [P, P, P, L, L, P, P, L, I, I, I, I, P]),
    "vlg_nchw_to_padded": (I, [P, P, I, I, I, I, I, I, P]),
    "vlg_padded_to_nchw": (I, [P, P, I, I, I, I, I, P]),
    "vlg_fill_coords": (I, [P, I, I, I, I, I, P]),
    "vlg_upsample2x_fwd": (I, [P, P, I, I, I, I, P]),
    "vlg_upsample2x_bwd": (I, [P, P, I, I, I, I, I, P]),
    "vlg_maxpool2x2": (I, [P, P, I, I, I, I, P]),
    "vlg_score1x1_relu": (I, [P, P, P, P, I, I, I, I, I, P]),
    "vlg_hed_head": (I, [P, P, P, P, P, P, P, P, I, I, I, P]),
    "vlg_maxpool2x2_bwd": (I, [P, P, P, I, I, I, I, P]),
    "vlg_l1_relu_padded": (I, [P, P, P, P, P, L, I, L, F, P]),
    "vlg_add_rows": (I, [P, P, L, I, P]),
    "vlg_sum_partials": (I, [P, I, P, I, P]),
}
# entry points of the DIAGNOSTIC build only (make -C csrc diag; VLG_HIP_LIB=.../libvlg_hip_diag.so): typed when present
DIAG_SIGNATURES = {
    "vlg_debug_set_clock_probe": (None, [P]),
    "vlg_debug_set_conv_probe": (None, [P]),
    "vlg_debug_set_gemm_bk": (None, [I]),
    "vlg_debug_set_gemm_run": (None, [I]),
}
CEPI_BIAS, CEPI_RESID, CEPI_PRELU, CEPI_DPRELU, CEPI_ACCUM, CEPI_CIN4 = 1, 2, 4, 8, 16, 32

_lib = None


class HipError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """dlopen libvlg_hip.so and type every entry point; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipError(
            "libvlg_hip.so not found at %s - run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C video-layout-generation_amd/csrc`); there is no CPU fallback" % LIB_PATH)
    # PyTorch-ROCm ships its own libamdhip64.so; it must be in the process BEFORE libvlg_hip.so is dlopen'ed so that both
    # bind to ONE HIP runtime (torch streams and device pointers are handed to the kernels).  Loaded the other way
    # round the library pulls /opt/rocm's runtime first and every launch fails with hipErrorNoDevice.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in DIAG_SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    return lib


def require_diag() -> ctypes.CDLL:
    """Development tools only: the diagnostic build of the library (vlg_debug_set_* switches)."""
    lib = load()
    if not hasattr(lib, "vlg_debug_set_clock_probe"):
        raise HipError("this tool needs the diagnostic build: `make -C video-layout-generation_amd/csrc diag` and run with "
                       "VLG_HIP_LIB=<repo>/video-layout-generation_amd/libvlg_hip_diag.so (the product library has no debug switches)")
    return lib


_ERR = {1001: "VLG_ERR_SHAPE (unsupported or inconsistent shape)", 1002: "VLG_ERR_ALIGN (pointer / leading dimension not 16-byte aligned)"}


def check(code: int, what: str) -> None:
    if code != 0:
        raise HipError("%s failed: %s" % (what, _ERR.get(code, "hipError_t %d" % code)))


def ptr(t) -> int:
    """Device pointer of a torch tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()


tracer = None     # optional callable(name, args) -> context manager | None: measurement hook (bench.py brackets launches with events)


def call(name: str, *args) -> None:
    cm = tracer(name, args) if tracer is not None else None
    if cm is None:
        check(getattr(load(), name)(*args), name)
        return
    with cm:
        check(getattr(load(), name)(*args), name)
