"""Build-time guard for the kernels that read LDS with inline-asm transposing loads (`frag_km_nowait`, gemm_bf16_dma.hip).

hipcc does not track an asm `ds_read_b64_tr_b16`: the kernel waits for it itself before the first MFMA that consumes it.  That is only
sound while the compiler does not TOUCH the result in between -- if register pressure makes it park a fragment in scratch (or copy it),
the copy is taken before the data has arrived (seen once while experimenting: wrong results, 9x slower; profiles/r02_gemm_refill_ab.jsonl).
So the device ISA of those kernels must use no scratch at all.  Runs on CPU (hipcc cross-compiles gfx950)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.timeout(300)
def test_kernels_with_untracked_lds_reads_use_no_scratch(tmp_path):
    out = tmp_path / "gemm_dma.s"
    src = os.path.join(ROOT, "internevo_amd", "csrc", "gemm_bf16_dma.hip")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.dirname(src), src, "-o", str(out)], check=True, capture_output=True)
    text = out.read_text()
    # kernel descriptors of the metadata block: name, private (scratch) segment size, spill counts
    kernels = re.findall(r"\.name:\s+(\S*gemm_dma_k\S*)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text)
    assert len(kernels) >= 40, f"metadata not parsed ({len(kernels)} kernels)"
    checked = 0
    for name, scratch, spills in kernels:
        m = re.search(r"ILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb([01])ELb([01])ELi?n?(\d+)E", name)
        assert m, name
        a_km, b_km = m.group(5) == "1", m.group(6) == "1"
        sched = -int(m.group(7)) if "ELin" in name else int(m.group(7))
        nowait = (sched in (-21, -22, -23) and (a_km or b_km)) or (sched == -4 and b_km and not a_km)
        if nowait:
            checked += 1
            assert int(scratch) == 0 and int(spills) == 0, f"{name}: {scratch} B of scratch, {spills} spilled VGPRs next to untracked LDS reads"
    assert checked >= 10, checked


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.timeout(600)
def test_dkdv_kernel_keeps_its_register_files(tmp_path):
    """The dK / dV attention kernel (flash_attn_bwd.hip) issues its MFMAs, its softmax and its LDS-DMA transfers from inline asm with explicit
    register files: accumulators and K / V fragments in AGPRs across the tile loop, the transfers unknown to hipcc.  What the build must keep:
    no scratch, no spills, no accumulator traffic between the register files inside the tile loop (hipcc once carried all 128 accumulator
    registers in arch VGPRs around the loop edge: 256 v_accvgpr moves per tile), and no compiler-inserted `s_waitcnt vmcnt` in the loop (the
    explicit ones sit in asm blocks): such a wait drains the transfers of the tiles ahead."""
    out = tmp_path / "fbwd.s"
    src = os.path.join(ROOT, "internevo_amd", "csrc", "flash_attn_bwd.hip")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-S", "--cuda-device-only", "-I",
                    os.path.join(ROOT, "include"), "-I", os.path.dirname(src), src, "-o", str(out)], check=True, capture_output=True)
    text = out.read_text()
    kernels = re.findall(r"\.name:\s+(\S*flash_dkdv_kI\S*)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text)
    assert len(kernels) >= 24, f"metadata not parsed ({len(kernels)} kernels)"
    for name, scratch, spills in kernels:
        assert int(scratch) == 0 and int(spills) == 0, f"{name}: {scratch} B of scratch, {spills} spilled VGPRs"
    checked = 0
    for m in re.finditer(r"^(_ZN\S*flash_dkdv_kI\w*):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        body = m.group(2)
        start = body.find("Loop Header")
        assert start > 0, m.group(1)
        end = max(mm.end() for mm in re.finditer(r"s_cbranch_scc[01] \.LBB", body))
        assert end > start, m.group(1)   # the back edge of the tile loop
        loop = body[start:end]
        assert loop.count("v_mfma_f32_32x32x16_bf16") >= 32, m.group(1)   # at least one whole tile
        # strip the inline-asm blocks: what is left is hipcc's own code
        own = re.sub(r";;#ASMSTART.*?;;#ASMEND", "", loop, flags=re.S)
        assert "v_accvgpr" not in own, f"{m.group(1)}: accumulator registers moved between the files inside the tile loop"
        assert "vmcnt" not in own, f"{m.group(1)}: compiler-inserted vmcnt wait inside the tile loop"
        checked += 1
    assert checked >= 24, checked
