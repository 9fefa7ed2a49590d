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


def test_scanner_follows_fall_through_into_a_labelled_block():
    """A writer at the end of one block and an asm reader at the top of the next: entered by fall-through there is no
    branch in between (advisor, round 4: the scanner used to forget every writer at a label)."""
    split = """
_Zkernel:
	v_dot4_u32_u8 v7, v49, s29, v7
	s_cbranch_scc1 .LBB0_9
.LBB0_2:
	;;#ASMSTART
	v_cvt_f32_ubyte2 v11, v7
	;;#ASMEND
"""
    found = isa_hazards.scan(split)
    assert len(found) == 1 and found[0][3] == "v_dot4_u32_u8" and found[0][1] == 1
    # the block above ends in an unconditional branch: this block is only ever reached by a taken branch
    assert isa_hazards.scan(split.replace("s_cbranch_scc1 .LBB0_9", "s_branch .LBB0_9")) == []
    # enough instructions in between, across the label
    assert isa_hazards.scan(split.replace(".LBB0_2:", "v_mov_b32 v1, v2\n.LBB0_2:\n\tv_mov_b32 v3, v4")) == []
    # a loop: the writer at the bottom, the reader at the head, the back edge a taken branch -- but the first entry falls in
    loop = """
_Zk2:
	v_mfma_f32_32x32x8_f16 v[0:15], v[20:21], v[22:23], v[0:15]
.LBB1_1:
	;;#ASMSTART
	v_add_f32 v30, v3, v31
	;;#ASMEND
	s_cbranch_vccnz .LBB1_1
"""
    assert len(isa_hazards.scan(loop)) == 1


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
