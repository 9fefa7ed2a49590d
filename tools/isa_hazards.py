#!/usr/bin/env python3
"""Scan gfx950 assembly (hipcc -S / -save-temps) for the one hazard the compiler cannot see for us: an inline-asm vector
instruction reading a VGPR that a v_dot* / v_mfma* instruction wrote fewer than `need` wait states earlier in the same basic
block (and, with one wait state, a transcendental instruction's result).  The hazard recogniser knows what the compiler's own instructions are and pads them; for an asm statement it only
knows the registers.  (Found the hard way: an asm v_mad_u32_u24 one wait state behind a v_dot4_u32_u8 read the old value
on a third of the pixels; three wait states, the distance LLVM keeps for its own non-DOT readers, is what the other asm
sites have.)

usage: python tools/isa_hazards.py file.s [...]   -> exit status 1 if any site is closer than 3 wait states
       scan(text, need=3) -> list of (function, distance, reader line, writer opcode)"""
import re
import sys

_TRANS = ("v_sqrt_", "v_rsq_", "v_rcp_", "v_exp_", "v_log_", "v_sin_", "v_cos_")
_REG = re.compile(r"^v(\d+)$")
_RANGE = re.compile(r"^v\[(\d+):(\d+)\]$")


def _regs(tok):
    tok = tok.strip().rstrip(",")
    m = _REG.match(tok)
    if m:
        return [int(m.group(1))]
    m = _RANGE.match(tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []


def scan(text, need=3):
    found = []
    fn, last, pos, in_asm, prev_op = None, {}, 0, False, None
    for ln in text.splitlines():
        s = ln.strip()
        if not s:
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if s.startswith(";"):
            continue
        if s.endswith(":") or (":" in s and s.split(":")[0].startswith((".LBB", "_Z"))):
            name = s.split(":")[0]
            if name.startswith(".LBB") or name.startswith(".L"):
                # A label.  A block entered by FALL-THROUGH (a loop header, the join behind an if) adds no cycles: a writer at
                # the end of the block above is as close to a reader at the top of this one as the instruction count says, so
                # the writer table is kept (advisor, round 4: it used to be dropped at every label -- the stale-VGPR fault seen in
                # blur.hip could have hidden behind one).  Only when the block above cannot fall through -- it ends in an
                # unconditional s_branch, s_setpc or s_endpgm -- is every way in a taken branch, which is more wait states
                # than any of these hazards needs, and the table starts empty.
                if prev_op in ("s_branch", "s_endpgm", "s_setpc_b64"):
                    last = {}
                continue
            if not name.startswith("."):
                fn, last, pos, prev_op = name, {}, 0, None
            continue
        if s.startswith("."):
            continue
        parts = s.split(None, 1)
        op = parts[0]
        args = parts[1].split(",") if len(parts) > 1 else []
        prev_op = op
        if op == "s_nop":
            pos += int(args[0].split(";")[0]) + 1
            continue
        pos += 1
        if not op.startswith("v_") or not args:
            continue
        if in_asm:
            for a in args[1:]:
                for r in _regs(a.split(";")[0]):
                    w = last.get(r)
                    if not w:
                        continue
                    d = pos - w[0] - 1
                    if w[1].startswith(("v_dot", "v_mfma", "v_smfmac")) and d < need:
                        found.append((fn, d, s, w[1]))
                    # gfx940+: a non-transcendental reader right behind a transcendental writer wants one wait state
                    if w[1].startswith(_TRANS) and d < 1:
                        found.append((fn, d, s, w[1]))
        for r in _regs(args[0]):
            last[r] = (pos, op)
    return found


if __name__ == "__main__":
    bad = 0
    for path in sys.argv[1:]:
        for fn, d, reader, writer in scan(open(path).read()):
            print(f"{path}: {fn}: `{reader}` reads a {writer} result {d} wait state(s) later")
            bad += 1
    sys.exit(1 if bad else 0)
