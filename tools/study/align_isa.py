#!/usr/bin/env python3
"""EXPERIMENT, NOT PART OF THE BUILD (round 3; result: 2.7 % SLOWER, profiles/r03_layout_experiment.txt).
Layout pass over the gfx950 assembly of the EM kernels, run between compiler and assembler by tools/study/build_aligned.sh.

Why it lost: the premise below came from a stream of nothing but 8-byte instructions.  csrc/tools/ubench_fetch2.hip / ubench_fetch3.hip
then showed that a misaligned 8-byte instruction costs nothing in mixed code (4-byte and 8-byte instructions alternating: 4.30 cycles per
instruction either way) and ~0.5 cycles only inside runs of six or more 8-byte instructions: what a lone wave runs into is the
instruction-fetch bandwidth (~8 bytes per issue slot), and this pass ADDS bytes (every re-encoding 4, every s_nop 4).
Kept as the record of the experiment.


Measured on MI355X (csrc/tools/ubench_fetch.hip, profiles/r03/ubench_fetch.txt): a lone wave issues an 8-byte
instruction (VOP3, DPP, DS ...) in 4.3 cycles when it starts on an 8-byte boundary and in 5.3 when it starts 4 bytes
behind one, and a taken branch costs 20 cycles to a target on a 32-byte boundary plus 2 per dword the target sits behind
it.  The compiler lays instructions out back to back, so behind every odd run of 4-byte instructions (SALU, the e32
forms of VOP1 / VOP2 / VOPC) all 8-byte instructions are misaligned until the next odd run: 40-50 % of the 190-200
8-byte instructions of an EM iteration, i.e. the "code placement" lottery of DESIGN.md section 4 (the same loop shifted
by 4 bytes ran up to 5 % slower or faster).

This pass removes the lottery instead of playing it:
  * an 8-byte instruction that would start 4 bytes behind an 8-byte boundary gets the nearest 4-byte VALU instruction
    in front of it (behind the previous 8-byte instruction) re-encoded in its VOP3 form (`_e32` -> `_e64`: same
    operation, same operands, 8 bytes) -- free of charge; where the run in front holds no such instruction (SALU
    only), an `s_nop 0` goes in if at least MIN_RUN 8-byte instructions follow back to back (it costs an issue slot);
  * the head of every loop (a label that a later branch jumps back to) is put on a 32-byte boundary.

Input: the assembly with encodings as `llvm-mc -show-encoding` prints it (sizes come from there, not from a table of
this script's own).  Output: assembly for the same assembler.
    align_isa.py in.enc.s out.s
"""
import re
import sys

import os
MIN_RUN = int(os.environ.get("ALIGN_NOP_MIN_RUN", "6"))            # s_nop only in front of at least this many 8-byte instructions (1000: never)
PROMOTE_MIN_RUN = int(os.environ.get("ALIGN_PROMOTE_MIN_RUN", "1"))  # re-encode only in front of at least this many (second experiment: 8)
ALIGN_HEADS = os.environ.get("ALIGN_HEADS", "1") == "1"
src, dst = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")

enc_re = re.compile(r";\s*encoding:\s*\[([^\]]*)\]")
label_re = re.compile(r"^([.\w$]+):")
branch_re = re.compile(r"^\s+s_(?:cbranch_\w+|branch)\s+([.\w$]+)")
p2align_re = re.compile(r"^\s+\.p2align\s+(\d+)")
# VALU instructions whose e32 and e64 forms take the same operand text on gfx9 / gfx950
promotable = re.compile(r"^\s+(v_(?:mov_b32|mov_b64|add_u32|sub_u32|subrev_u32|and_b32|or_b32|xor_b32|lshlrev_b32|lshrrev_b32|ashrrev_i32|"
                        r"cndmask_b32|fmac_f64|add_f32|mul_f32|max_u32|min_u32|max_i32|min_i32|mul_u32_u24|mul_i32_i24|lshlrev_b64|"
                        r"cmp_\w+?|cvt_\w+?|not_b32|bfrev_b32|add_co_u32|addc_co_u32|sub_co_u32|subb_co_u32|ldexp_f64|rcp_f64|"
                        r"fract_f64|floor_f64|trunc_f64|rndne_f64|sqrt_f64|rsq_f64|frexp_mant_f64|frexp_exp_i32_f64))_e32\b")

# labels that some later branch jumps back to: loop heads
label_line = {}
for i, l in enumerate(lines):
    m = label_re.match(l)
    if m:
        label_line[m.group(1)] = i
loop_heads = set()
for i, l in enumerate(lines):
    m = branch_re.match(l)
    if m and m.group(1) in label_line and label_line[m.group(1)] < i:
        loop_heads.add(label_line[m.group(1)])


def size_of(l):
    m = enc_re.search(l)
    return len(m.group(1).split(",")) if m else 0


out = []
off = 0            # byte offset modulo 256 from the last boundary we know (function starts are .p2align 8)
run4 = []          # indices into `out` of the 4-byte instructions since the last 8-byte one
stats = dict(eight=0, misaligned=0, promoted=0, nops=0, left=0, heads=0)
in_text = False
n = len(lines)
for i, l in enumerate(lines):
    if l.startswith("\t.section") or l.startswith("\t.text"):
        in_text = ".text" in l
        off, run4 = 0, []
        out.append(l)
        continue
    if not in_text:
        out.append(l)
        continue
    m = p2align_re.match(l)
    if m:
        a = 1 << int(m.group(1))
        off = (off + a - 1) // a * a if a <= 256 else 0
        run4 = []
        out.append(l)
        continue
    if ALIGN_HEADS and i in loop_heads and off % 32 != 0:
        out.append("\t.p2align 5")
        off = (off + 31) // 32 * 32
        run4 = []
        stats["heads"] += 1
    sz = size_of(l)
    if sz == 0:
        out.append(l)
        continue
    if sz == 4:
        run4.append(len(out))
        out.append(l)
        off += 4
        continue
    if sz % 8 == 0:
        stats["eight"] += 1
        if off % 8 == 4:
            stats["misaligned"] += 1
            fixed = False
            follow0 = 0  # 8-byte instructions from here on, back to back
            for j in range(i, n):
                s_ = size_of(lines[j])
                if s_ == 0 and not label_re.match(lines[j]) and not lines[j].strip().startswith("."):
                    continue
                if s_ % 8 == 0 and s_ > 0:
                    follow0 += 1
                else:
                    break
            for k in (reversed(run4) if follow0 >= PROMOTE_MIN_RUN else []):
                if promotable.match(out[k]):
                    out[k] = enc_re.sub("; (re-encoded as VOP3 by align_isa.py)", out[k].replace("_e32", "_e64", 1))
                    off += 4
                    stats["promoted"] += 1
                    fixed = True
                    break
            if not fixed:
                follow = 0  # 8-byte instructions from here on, back to back
                for j in range(i, n):
                    s = size_of(lines[j])
                    if s == 0 and not label_re.match(lines[j]) and not lines[j].strip().startswith("."):
                        continue
                    if s % 8 == 0 and s > 0:
                        follow += 1
                    else:
                        break
                if follow >= MIN_RUN:
                    out.append("\ts_nop 0 ; (align_isa.py)")
                    off += 4
                    stats["nops"] += 1
                else:
                    stats["left"] += 1
        out.append(l)
        off += sz
        run4 = []
        continue
    # anything else (12-byte encodings do not exist on gfx950; keep the offset honest anyway)
    out.append(l)
    off += sz
    run4 = []
open(dst, "w").write("\n".join(out))
print("align_isa: %(eight)d 8-byte instructions, %(misaligned)d would start 4 bytes behind an 8-byte boundary: %(promoted)d fixed by a VOP3 "
      "re-encoding in front, %(nops)d by an s_nop, %(left)d left; %(heads)d loop heads moved to 32-byte boundaries" % stats, file=sys.stderr)
