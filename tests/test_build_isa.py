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
