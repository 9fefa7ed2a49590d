"""The kernels that use inline-asm vector instructions are compiled to assembly and scanned for an asm statement reading a
v_dot* / v_mfma* result too early (tools/isa_hazards.py): the compiler pads its own instructions, not ours.  CPU only
(hipcc cross-compiles); the scanner itself is checked on the sequence that failed on the GPU."""
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_hazards  # noqa: E402


def test_scanner_flags_the_sequence_that_failed():
    bad = """
_Zkernel:
	v_dot4_u32_u8 v7, v49, s29, v7
	s_nop 0
	;;#ASMSTART
	v_mad_u32_u24 v11, v7, v34, v35
	;;#ASMEND
"""
    found = isa_hazards.scan(bad)
    assert len(found) == 1 and found[0][1] == 1 and found[0][3] == "v_dot4_u32_u8"
    ok = bad.replace("s_nop 0", "s_nop 2")
    assert isa_hazards.scan(ok) == []
    trans = """
_Zk:
	v_sqrt_f32_e32 v3, v2
	;;#ASMSTART
	v_max3_f32 v4, v3, v1, v0
	;;#ASMEND
"""
    assert len(isa_hazards.scan(trans)) == 1                       # a transcendental's result, read at once
    assert isa_hazards.scan(trans.replace(";;#ASMSTART", "v_mov_b32 v9, v8\n\t;;#ASMSTART")) == []
    # a compiler-selected reader is the compiler's business
    assert isa_hazards.scan(bad.replace(";;#ASMSTART", "").replace(";;#ASMEND", "")) == []


def sources_with_asm_instructions():
    from vslam_amd import build
    out = []
    for src in build.SOURCES:
        text = open(os.path.join(build.CSRC, src)).read()
        hdrs = [h for h in ("image_common.h", "ctx.h", "introselect.h") if f'"{h}"' in text]
        text += "".join(open(os.path.join(build.CSRC, h)).read() for h in hdrs)
        import re
        used = set(re.findall(r'asm(?:\s+volatile)?\s*\(\s*"(v_[a-z0-9_]+)', text))
        if used:
            out.append(src)
    return out


def test_no_asm_statement_reads_a_dot_result_too_early():
    from vslam_amd import build
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = sources_with_asm_instructions()
    assert "response.hip" in srcs
    with tempfile.TemporaryDirectory() as tmp:
        def compile_s(src):
            out = os.path.join(tmp, src.replace(".hip", ".s"))
            cmd = [hipcc] + build.FLAGS + build.EXTRA_FLAGS.get(src, []) + ["--cuda-device-only", "-S", "-o", out, src]
            subprocess.run(cmd, cwd=build.CSRC, check=True, stderr=subprocess.DEVNULL)
            return out
        with ThreadPoolExecutor(max_workers=4) as pool:
            outs = list(pool.map(compile_s, srcs))
        found = []
        for path in outs:
            text = open(path).read()
            if path.endswith("response.s"):
                assert "ASMSTART" in text   # the scanner has something to look at
            found += [(os.path.basename(path),) + f for f in isa_hazards.scan(text)]
    assert not found, found
