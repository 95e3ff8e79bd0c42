"""Static checks on the gfx950 assembly of the kernels (hipcc cross-compiles without a GPU)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="needs hipcc")


@pytest.fixture(scope="module")
def kernel_asm(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("asm") / "pv_kernels.s")
    flags = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize"]  # the Makefile's
    cmd = [HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", *flags, f"-I{ROOT}/include",
           f"-I{ROOT}/audiomod_amd/csrc", "--cuda-device-only", "-S", f"{ROOT}/audiomod_amd/csrc/pv_kernels.hip", "-o", out]
    subprocess.run(cmd, check=True, capture_output=True)
    return open(out).read().split("\n")


def test_inline_asm_selects_keep_their_distance_from_sgpr_writers(kernel_asm):
    """The fused kernel zeroes finalised accumulator samples with `v_cndmask_b32_e64 ... s[mask]` written as inline
    assembly (pv_kernels.hip chain_slice_tail, tagged `pvmask`).  On gfx950 a VALU instruction reading an SGPR that a
    VALU instruction wrote (v_cmp, v_readlane, ...) needs two wait states in between, and the compiler's hazard
    recogniser does not look inside inline assembly -- the masks are therefore made and pinned in front of the turn's
    wait loop.  This checks the generated code: no VALU writer of a mask register within four instructions of its
    inline-assembly reader."""
    hits = bad = 0
    for i, l in enumerate(kernel_asm):
        if "pvmask" not in l:
            continue
        m = re.search(r"v_cndmask_b32_e64 v\d+, v\d+, 0, s\[(\d+):(\d+)\]", l)
        assert m, l
        hits += 1
        lo, hi = int(m.group(1)), int(m.group(2))
        k, seen = i - 1, 0
        while seen < 4 and k > 0:
            s = kernel_asm[k].strip()
            k -= 1
            if not s or s.startswith(";") or s.startswith(".") or "pvmask" in s:
                continue
            seen += 1
            if s.startswith("v_"):
                dst = ",".join(s.split(None, 1)[1].split(",")[0:2])
                if re.search(r"\bs%d\b|\bs%d\b|s\[%d:%d\]" % (lo, hi, lo, hi), dst):
                    bad += 1
    assert hits > 0 and bad == 0, (hits, bad)


def test_wave_analysis_kernels_have_no_static_lds_or_scratch(kernel_asm):
    """atan2f's interval table is addressed by a compile-time LDS address (dynamic LDS must start at 0: no static LDS in
    those kernels; the engine checks the same at run time), and the hot kernels of the bench path must not spill."""
    text = "\n".join(kernel_asm)
    for name in ("_ZN2pv22pv_analyze_wave_kernelILi1024ELi1EEEvNS_11AnalyzeArgsE",
                 "_ZN2pv22pv_analyze_wave_kernelILi2048ELi1EEEvNS_11AnalyzeArgsE"):
        i = text.index("\n" + name + ":")
        blk = text[i:text.index(".end_amdhsa_kernel", i)]
        assert re.search(r"\.amdhsa_group_segment_fixed_size 0\b", blk), name
        assert re.search(r"\.set %s\.private_seg_size, 0\b" % re.escape(name), text), name
