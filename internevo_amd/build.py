"""Build libinternevo_hip.so (and the oracle's C pieces, if any) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU.  Objects are cached by source mtime so rebuilding after a
one-file edit takes seconds.  The .so is git-ignored but travels to the GPU box with the snapshot.
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(CSRC, "libinternevo_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wno-unused-result"]
# per-source additions.  The flash kernels place their vector-ALU work by hand in the shadow of single MFMAs; SLP-packed f32
# arithmetic (v_pk_add_f32 / v_pk_mul_f32) costs more issue time there than the two scalar instructions it replaces
# (MI355X_MICROARCH.md, "price of one filler beside MFMAs").
EXTRA_FLAGS = {"flash_attn_fwd.hip": ["-fno-slp-vectorize"], "flash_attn_bwd.hip": ["-fno-slp-vectorize"]}
KBENCH_SRC = os.path.join(HERE, "..", "tools", "kbench", "kbench.cpp")
KBENCH_BIN = os.path.join(HERE, "..", "tools", "kbench", "kbench")


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm's hipcc to build libinternevo_hip.so)")


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "internevo_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, force):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    spath = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(spath), _headers_mtime()):
        return obj, False
    cmd = [_hipcc(), *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", spath, "-o", obj]
    if os.path.exists(obj):
        os.remove(obj)  # a stale object must never survive a failed compile (the hipcc wrapper can exit 0 after "failed to execute")
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0 or not os.path.exists(obj) or "error:" in res.stderr:
        raise RuntimeError(f"hipcc failed for {src}:\n{res.stdout}\n{res.stderr}")
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in results]
    rebuilt = any(r for _, r in results)
    if rebuilt or force or not os.path.exists(LIB):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
    if verbose:
        print(f"[internevo_amd.build] {LIB} ({'rebuilt' if rebuilt else 'up to date'}; {len(srcs)} HIP sources)")
    return LIB


def build_kbench(verbose=True):
    """tools/kbench: the torch-free A/B harness (development tool; dlopen()s the library)."""
    if os.path.exists(KBENCH_BIN) and os.path.getmtime(KBENCH_BIN) > os.path.getmtime(KBENCH_SRC):
        return KBENCH_BIN
    cmd = [_hipcc(), "-O2", "-std=c++17", f"--offload-arch={ARCH}", KBENCH_SRC, "-o", KBENCH_BIN, "-ldl"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0 or not os.path.exists(KBENCH_BIN):
        raise RuntimeError(f"kbench build failed:\n{res.stdout}\n{res.stderr}")
    if verbose:
        print(f"[internevo_amd.build] {KBENCH_BIN}")
    return KBENCH_BIN


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    if "--kbench" in sys.argv:
        build_kbench()
