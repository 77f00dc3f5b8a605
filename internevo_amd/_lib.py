"""ctypes binding of libinternevo_hip.so (the C ABI declared in include/internevo_hip.h).

The library is the product: there is NO fallback.  If the shared object is missing or an entry point
cannot be resolved, importing the kernels fails loudly (``InternEvoHipError``), and every non-zero
return code of the C ABI is raised as ``InternEvoHipError`` carrying ``ie_last_error()``.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libinternevo_hip.so")

IE_BF16 = 0
IE_F32 = 1
ABI_VERSION = 1


class InternEvoHipError(RuntimeError):
    pass


class IeStepState(Structure):
    _fields_ = [
        ("loss_scale", c_float),
        ("growth_step", c_int),
        ("hysteresis_step", c_int),
        ("adam_step", c_int),
        ("skip", c_int),
        ("found_inf", c_int),
        ("found_nan", c_int),
        ("inv_scale", c_float),
        ("grad_norm", c_float),
        ("loss_scale_used", c_float),
        ("skipped_total", c_int),
        ("_pad", c_int),
    ]


class IeScalerConfig(Structure):
    _fields_ = [
        ("growth_factor", c_float),
        ("backoff_factor", c_float),
        ("min_scale", c_float),
        ("max_scale", c_float),
        ("growth_interval", c_int),
        ("hysteresis", c_int),
        ("clip_grad_norm", c_float),
        ("dynamic", c_int),
    ]


P = c_void_p
I64 = c_int64
I = c_int
F = c_float
Dbl = c_double

# name -> (restype, argtypes).  Must list EVERY symbol of include/internevo_hip.h
# (tests/test_abi.py cross-checks this table against the header).
SIGNATURES = {
    "ie_abi_version": (I, []),
    "ie_last_error": (c_char_p, []),
    "ie_rmsnorm_fwd": (I, [P, I, P, I, P, P, I64, I64, F, P]),
    "ie_add_rmsnorm_fwd": (I, [P, P, P, P, P, P, I64, I64, F, P]),
    "ie_rmsnorm_bwd_partials": (I64, [I64]),
    "ie_rmsnorm_bwd": (I, [P, P, I, P, I, P, P, P, P, P, I, I64, I64, P]),
    "ie_apply_rotary": (I, [P, P, P, P, P, P, I, I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, I, P]),
    "ie_qkv_rotary_fwd": (I, [P, P, P, P, P, P, I64, I, I, I, I, P]),
    "ie_qkv_rotary_bwd": (I, [P, P, P, P, P, P, I64, I, I, I, I, P]),
    "ie_qkv_rotary_fwd_scaled": (I, [P, P, P, P, P, P, I64, I, I, I, I, F, P]),
    "ie_qkv_rotary_bwd_scaled": (I, [P, P, P, P, P, P, I64, I, I, I, I, F, P]),
    "ie_swiglu_fwd": (I, [P, I64, P, I64, P, I64, I64, I64, P]),
    "ie_swiglu_bwd": (I, [P, I64, P, I64, P, I64, P, I64, P, I64, P, I64, I64, I64, P]),
    "ie_ce_fwd": (I, [P, I, I64, P, P, P, I64, I64, I64, F, P]),
    "ie_ce_mean": (I, [P, P, I64, I64, P, P, P]),
    "ie_ce_fwd_metric": (I, [P, I, I64, P, P, P, P, P, I64, I64, I64, F, P]),
    "ie_metric_accumulate": (I, [P, P, P, P, I64, I64, I, P, P, P, P, P, P]),
    "ie_ce_bwd": (I, [P, P, I, I64, P, P, P, F, P, I64, I64, I64, F, P]),
    "ie_sumsq_max_partials": (I64, []),
    "ie_sumsq_partial": (I, [P, I, I64, P, I64, POINTER(c_int64), P]),
    "ie_sumsq_finish": (I, [P, I64, P, I, P]),
    "ie_step_state_init": (I, [P, F, P]),
    "ie_step_control": (I, [P, P, POINTER(IeScalerConfig), P]),
    "ie_adamw_step": (I, [P, I, P, P, P, P, I64, P, Dbl, Dbl, Dbl, Dbl, Dbl, P]),
    "ie_embedding_fwd": (I, [P, P, P, I64, I64, I64, P]),
    "ie_embedding_bwd": (I, [P, P, P, P, I64, I64, I64, I, P]),
    "ie_add_bf16": (I, [P, P, P, I64, P]),
    "ie_seq_head_permute": (I, [P, P, I64, I, I, I64, I, P]),
    "ie_scale_bf16": (I, [P, I64, F, P]),
    "ie_grad_scale_mix": (I, [P, I64, F, P]),
    "ie_head_weight_fwd": (I, [P, I64, P, I64, P, I64, I64, F, I, P]),
    "ie_head_weight_bwd": (I, [P, I64, P, I64, P, P, I64, I64, I64, F, I, I, P]),
    "ie_cast": (I, [P, I, P, I, I64, P]),
    "ie_gemm_bf16": (I, [P, I64, I, P, I64, I, P, I64, I64, I64, I64, I, P]),
    "ie_gemm_bf16_tile": (I, [I, P, I64, I, P, I64, I, P, I64, I64, I64, I64, I, P]),
    "ie_fp8_amax": (I, [P, I64, I64, P, P]),
    "ie_fp8_quantize": (I, [P, I64, I64, P, P, P, P]),
    "ie_gemm_fp8": (I, [P, I64, P, I64, P, I64, I64, I64, I64, P, P, I, P]),
    "ie_gemm_fp8_batched": (I, [P, I64, I64, P, I64, I64, P, I64, I64, I64, I64, I64, I64, P, P, I, P]),
    "ie_gemm_bf16_batched": (I, [P, I64, I64, I, P, I64, I64, I, P, I64, I64, I64, I64, I64, I, I, P]),
    "ie_colsum_bf16": (I, [P, I64, P, I64, I64, P]),
    "ie_flash_attn_fwd": (I, [P, I64, P, P, I64, P, I64, P, P, I, I64, I, I, I, I, F, I, P]),
    "ie_flash_attn_bwd": (I, [P, I64, P, I64, P, P, I64, P, I64, P, P, P, I64, P, P, I64, P, I, I64, I, I, I, I, F, I, P]),
    "ie_flash_attn_bwd_qkv_rotary_is_fused": (I, [I, I, I, I, I, I]),
    "ie_flash_attn_bwd_qkv_rotary": (I, [P, I64, P, I64, P, P, I64, P, I64, P, P, P, P, P, P, P, I, I64, I, I, I, I, F, I, P]),
    "ie_tune_flash_dq_occupancy": (I, [I]),
    "ie_tune_gemm_group": (I, [I]),
    "ie_tune_ffn_fuse": (I, [I]),
    "ie_gemm_qkv_rotary_fwd": (I, [P, I64, P, I64, P, P, P, P, P, P, I64, I64, I, I, I, I64, I, F, P]),
    "ie_tune_qkv_rotary_fuse": (I, [I]),
    "ie_gemm_qkv_rotary_is_fused": (I, [I64, I, I, I, I64]),
    "ie_gemm_swiglu_is_fused": (I, [I, I64, I64, I64]),
    "ie_gemm_swiglu_fwd": (I, [P, I64, P, I64, P, I64, P, I64, I64, I64, I64, P]),
    "ie_gemm_swiglu_bwd": (I, [P, I64, P, I64, P, I64, P, I64, P, I64, I64, I64, I64, P]),
    "ie_tune_gemm_tail_split": (I, [I]),
    "ie_tune_gemm_persistent": (I, [I]),
    "ie_tune_gemm_persistent_skip_n": (I, [I64]),
    "ie_tune_gemm_dgrad_refill_all": (I, [I]),
    "ie_tune_gemm_queue_memset": (I, [I]),
    "ie_linear_fwd_add": (I, [P, I64, P, I64, P, P, I64, I64, I64, I64, P]),
    "ie_tune_adamw_cus": (I, [I]),
    "ie_gemm_set_wgrad_ksplit_workspace": (I, [P, I64]),
    "ie_gemm_dma_persistent_takes": (I, [I64, I64, I64]),
    "ie_hold_cus": (I, [I, I, P]),
    "ie_gemm_dma_set_persistent_grid": (I, [I]),
    "ie_tune_flash_dkdv_split": (I, [I]),
    "ie_bias_add_bf16": (I, [P, I64, P, I64, I64, P]),
    "ie_step_control_groups": (I, [P, P, I, POINTER(IeScalerConfig), P, P, P]),
    "ie_adamw_step_group": (I, [P, I, P, P, P, P, I64, P, P, Dbl, Dbl, Dbl, Dbl, Dbl, P]),
    "ie_moe_gumbel_noise": (I, [P, I64, ctypes.c_uint32, ctypes.c_uint64, P]),
    "ie_moe_gate_fwd": (I, [P, I64, P, P, I64, I, I, P, P, P, P]),
    "ie_moe_route": (I, [P, P, I64, I, I, P, P, P, P, P, P]),
    "ie_moe_chunk_rows": (I, [P, P, P, I64, I, I, I, P]),
    "ie_moe_dispatch": (I, [P, I64, P, I64, I, P, P]),
    "ie_moe_combine_fwd": (I, [P, P, P, I64, I, P, I64, P]),
    "ie_moe_combine_bwd": (I, [P, I64, P, P, P, I64, I64, I, P, P, P]),
    "ie_moe_dispatch_bwd": (I, [P, P, P, I64, I, P, I64, P]),
    "ie_moe_dwg_workspace": (I64, [I, I]),
    "ie_moe_gate_bwd": (I, [P, I64, P, P, P, P, P, P, P, F, I64, I, I, P, P, I64, P, I, P, P]),
    "ie_gemm_last_kernel": (I, [c_char_p, I]),
    "ie_gemm_note_kernel": (I, [I, I, I, I, I, I, I, I, I]),
    "ie_tune_flash_fwd_variant": (I, [I]),
    "ie_tune_flash_bwd_variant": (I, [I]),
    "ie_flash_attn_fwd_x": (I, [P, I64, P, P, I64, P, I64, P, P, P, I, I64, I64, I, I, I, I, F, P]),
    "ie_flash_attn_bwd_x": (I, [P, I64, P, I64, P, P, I64, P, I64, P, P, P, I64, P, P, I64, P, P, I, I64, I64, I, I, I, I, I, F, P]),
    "ie_attn_merge": (I, [P, P, I64, P, I64, P, I64, I64, I, I, P]),
    "ie_acc_bf16": (I, [P, P, I64, P]),
    "ie_flash_attn_bwd_spill_bytes": (I64, [I, I, I, I]),
    "ie_flash_attn_bwd_set_spill": (I, [P, I64]),
    "ie_flash_attn_bwd_workspace": (I64, [I64, I, I, I]),
    "ie_mfma_probe": (I, [P, P, P, P]),
}

_lib = None


def load():
    """Load the shared library (once) and bind every entry point.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise InternEvoHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU or PyTorch fallback for the hot path."
        )
    # torch ships its own libamdhip64.so; it must be the HIP runtime already resident in the process when our
    # library is loaded, otherwise two runtimes coexist and torch's streams/pointers are foreign to our launches
    # (observed on MI355X: hipErrorInvalid* from the first launch when the .so was loaded before `import torch`).
    import torch  # noqa: F401

    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the ROCm runtime being present
        raise InternEvoHipError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise InternEvoHipError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    ver = lib.ie_abi_version()
    if ver != ABI_VERSION:
        raise InternEvoHipError(f"ABI version mismatch: library {ver}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().ie_last_error()
        raise InternEvoHipError(f"{what or 'libinternevo_hip'} failed (code {rc}): {msg.decode() if msg else ''}")
