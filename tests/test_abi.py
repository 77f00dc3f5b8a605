"""CPU checks of the drop-in boundary: the C-ABI library builds, loads, and exports exactly the symbols
include/internevo_hip.h declares; the ctypes table matches the header; no compute is launched."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "internevo_hip.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"typedef struct.*?}\s*\w+;", "", src, flags=re.S)
    decls = re.findall(r"\b(?:int|int64_t|const char\s*\*)\s+(ie_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S)
    return {name: args for name, args in decls}


@pytest.fixture(scope="module")
def lib():
    from internevo_amd import _lib
    from internevo_amd.build import build

    build(verbose=False)  # hipcc cross-compiles gfx950 without a GPU
    return _lib.load()


def test_header_and_binding_agree(lib):
    from internevo_amd import _lib

    hdr = header_functions()
    assert len(hdr) >= 25
    assert set(hdr) == set(_lib.SIGNATURES), (set(hdr) ^ set(_lib.SIGNATURES))
    for name, args in hdr.items():
        nargs = 0 if args.strip() in ("", "void") else len([a for a in args.split(",") if a.strip()])
        assert nargs == len(_lib.SIGNATURES[name][1]), f"{name}: header has {nargs} args, binding {len(_lib.SIGNATURES[name][1])}"


def test_library_exports_every_declared_symbol(lib):
    from internevo_amd import _lib

    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    for name in header_functions():
        assert name in exported, f"{name} declared in the header but not exported by the .so"
    assert lib.ie_abi_version() == _lib.ABI_VERSION


def test_argument_errors_are_reported_without_a_gpu(lib):
    from internevo_amd import _lib

    # null pointers are rejected before any HIP call
    rc = lib.ie_rmsnorm_fwd(None, 0, None, 0, None, None, 4, 4, 1e-5, None)
    assert rc == -1 and b"null" in lib.ie_last_error()
    rc = lib.ie_gemm_bf16(ctypes.c_void_p(16), 7, 0, ctypes.c_void_p(16), 8, 0, ctypes.c_void_p(16), 8, 8, 8, 8, 0, None)
    assert rc in (-1, -2)
    with pytest.raises(_lib.InternEvoHipError):
        _lib.check(rc, "ie_gemm_bf16")


def test_round6_host_side_decisions_without_a_gpu(lib):
    """The host-only entry points of round 6 (no launch, no device): which shapes the fused attention backward and the residual-add epilogue take, and the
    ranges of the new tuning hooks."""
    fused = lib.ie_flash_attn_bwd_qkv_rotary_is_fused
    assert fused(4, 4096, 32, 8, 128, 1) == 1          # the benchmark's call: enough key blocks for no head split
    assert fused(3, 700, 6, 2, 128, 1) == 1            # an odd GQA group cannot be split
    assert fused(2, 512, 4, 2, 128, 1) == 0            # a small problem: the dK / dV kernel splits the heads and sums partials
    assert fused(4, 4096, 32, 8, 64, 1) == 0 and fused(4, 4096, 32, 8, 128, 0) == 0 and fused(0, 4096, 32, 8, 128, 1) == 0
    try:
        assert lib.ie_tune_flash_bwd_variant(4) == 0 and fused(4, 4096, 32, 8, 128, 1) == 0   # delta by its own kernel: the fused call steps aside
    finally:
        assert lib.ie_tune_flash_bwd_variant(0) == 0
    assert lib.ie_tune_flash_bwd_variant(8) != 0
    takes = lib.ie_gemm_dma_persistent_takes
    assert takes(16384, 4096, 4096) == 1 and takes(16384, 4096, 14336) == 1      # wo, w2 of the 7B step
    assert takes(4096, 4096, 1024) == 0 and takes(16384, 4096, 6144) == 0 and takes(1000, 4096, 4096) == 0   # one round; the frame's K rule; ragged rows
    assert lib.ie_tune_adamw_cus(-1) != 0 and lib.ie_tune_adamw_cus(257) != 0
    assert lib.ie_tune_adamw_cus(128) == 0 and lib.ie_tune_adamw_cus(0) == 0
    assert lib.ie_tune_gemm_queue_memset(2) != 0 and lib.ie_tune_gemm_queue_memset(0) == 0
    # a bad call of the fused backward is refused before anything is launched
    assert lib.ie_flash_attn_bwd_qkv_rotary(None, 0, None, 0, None, None, 0, None, 0, None, None, None, None, None, None, None, 1, 16, 16, 4, 2, 128, 1.0, 1, None) != 0


def test_struct_layouts_match_header():
    from internevo_amd import _lib

    assert ctypes.sizeof(_lib.IeStepState) == 48
    assert ctypes.sizeof(_lib.IeScalerConfig) == 32
    assert _lib.IeStepState.loss_scale.offset == 0 and _lib.IeStepState.inv_scale.offset == 28


def test_kernels_refuse_cpu_tensors():
    import torch

    from internevo_amd import kernels

    with pytest.raises(ValueError):
        kernels.rmsnorm_fwd(torch.zeros(4, 512, dtype=torch.bfloat16), torch.ones(512, dtype=torch.bfloat16), 1e-5)


def test_product_does_not_import_the_oracle():
    """The product path must never route through the oracle (only tests / smoke / bench's cpu_baseline may)."""
    pkg = os.path.join(ROOT, "internevo_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
