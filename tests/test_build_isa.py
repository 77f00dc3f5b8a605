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
        nowait = (sched in (-21, -22, -23, -24) and (a_km or b_km)) or (sched in (-4, -5) and b_km and not a_km)
        if nowait:
            checked += 1
            assert int(scratch) == 0 and int(spills) == 0, f"{name}: {scratch} B of scratch, {spills} spilled VGPRs next to untracked LDS reads"
    assert checked >= 10, checked
    # the persistent frame (gemm_p5_k<B_KM, EPI>): its k-major instantiations read B with the same untracked loads -- plain and with the SwiGLU backward in
    # the epilogue (round 6: that epilogue first overflowed the register file and parked the fragment ADDRESSES of the k-loop in scratch)
    p5 = re.findall(r"\.name:\s+(\S*gemm_p5_kILb1E\S*)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text)
    assert len(p5) >= 2, p5
    for name, scratch, spills in p5:
        assert int(scratch) == 0 and int(spills) == 0, f"{name}: {scratch} B of scratch, {spills} spilled VGPRs next to untracked LDS reads"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.timeout(300)
def test_round4_gemm_schedules_keep_their_accumulators_in_agprs(tmp_path):
    """The schedules added in round 4 -- the refill schedule on v_mfma_f32_16x16x32_bf16 (-5), its ring form (-24) and the e4m3 schedule on
    v_mfma_f32_32x32x64_f8f6f4 (-6) -- use all 256 AGPRs for accumulators and nearly all VGPRs for fragments.  Twice while writing them hipcc answered a harmless
    source change with a very different allocation: with the 16x16x32 builtin it parked accumulator tiles in VGPRs and copied them around their MFMAs (120 to 480
    v_accvgpr moves per k-tile), and with the MFMA inside `if (!(m & 1))` it put accumulators AND fragments into scratch (1.7 KB per lane).  What the build must keep:
    no scratch, no spills, and a steady-state k-tile loop with the full count of MFMAs, 32 LDS fragment reads, 16 DMA pieces (per two entries for the ring) and no
    accumulator traffic between the register files."""
    out = tmp_path / "gemm_dma.s"
    src = os.path.join(ROOT, "internevo_amd", "csrc", "gemm_bf16_dma.hip")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.dirname(src), src, "-o", str(out)], check=True, capture_output=True)
    text = out.read_text()
    want = {"Lb0ELb0ELin5ELi0E": ("v_mfma_f32_16x16x32_bf16", 128, 32), "Lb0ELb0ELin5ELi1E": ("v_mfma_f32_16x16x32_bf16", 128, 32),
            "Lb0ELb1ELin5ELi0E": ("v_mfma_f32_16x16x32_bf16", 128, 48), "Lb1ELb1ELin24ELi0E": ("v_mfma_f32_16x16x32_bf16", 128, 64),
            "Lb0ELb0ELin6ELi0E": ("v_mfma_f32_32x32x64_f8f6f4", 32, 32)}
    seen = set()
    for m in re.finditer(r"^(_ZN\S*gemm_dma_kI\w*):[^\n]*\n(.*?)\.end_amdhsa_kernel", text, re.S | re.M):
        tag = next((t for t in want if t in m.group(1)), None)
        if tag is None:
            continue
        mfma, n_mfma, n_reads = want[tag]
        body = m.group(2)
        assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", body), f"{m.group(1)}: scratch in use"
        # the steady-state loop: the basic block with the most MFMAs
        blocks = re.split(r"\n\.LBB\d+_\d+:", body)
        loop = max(blocks, key=lambda b: b.count(mfma))
        assert loop.count(mfma) == n_mfma, f"{m.group(1)}: {loop.count(mfma)} MFMAs in the k-tile loop, expected {n_mfma}"
        assert len(re.findall(r"\bds_read_b(?:128|64_tr_b16)\b", loop)) == n_reads and loop.count("buffer_load_dwordx4") == 16, m.group(1)
        own = re.sub(r";;#ASMSTART.*?;;#ASMEND", "", loop, flags=re.S)
        assert "v_accvgpr" not in own and "scratch_" not in own, f"{m.group(1)}: accumulator copies / scratch traffic inside the k-tile loop"
        seen.add(tag)
    assert seen == set(want), sorted(set(want) - seen)


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


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.timeout(300)
def test_workgroups_that_must_sit_alone_on_a_cu_really_allocate_their_lds(tmp_path):
    """hold_cu_k (capi.hip) and adamw_cus_k (loss_optim.hip) pin one workgroup per CU with 96 KB of LDS nobody reads.  The compiler removes an array whose only
    store is dead (round 6 found hold_cu_k built with 0 bytes: the array has to be volatile): the code object's group segment size is the guarantee."""
    for fname, kernel in (("capi.hip", "hold_cu_k"), ("loss_optim.hip", "adamw_cus_k")):
        out = tmp_path / (fname + ".s")
        src = os.path.join(ROOT, "internevo_amd", "csrc", fname)
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"),
                        "-I", os.path.dirname(src), src, "-o", str(out)], check=True, capture_output=True)
        found = re.findall(r"\.group_segment_fixed_size:\s+(\d+)\n(?:(?!.*group_segment_fixed_size).*\n)*?\s+\.name:\s+(\S*" + kernel + r"\S*)\n", out.read_text())
        assert found, f"{kernel}: metadata not parsed"
        for lds, name in found:
            assert 96 * 1024 <= int(lds) <= 160 * 1024, f"{name}: {lds} B of LDS -- more than one such workgroup fits on a CU"
