#!/usr/bin/env python3
"""ISA of the EM loop of one kernel instantiation, cut into the phases of an iteration (at its s_barrier instructions)
with instruction-class counts per phase: the listing committed as profiles/r02_isa_loop_<name>.txt.

    python tools/isa_listing.py 'em_kernelILi0ELi1ELi2ELb0' > profiles/r02_isa_loop_e23_latency_ilp.txt

Builds colate_amd/csrc/em_kernels_ilp.hip with the Makefile's flags to a device ELF and disassembles it
(llvm-objdump).  Static counts over ALL paths of a phase (both roles, cold paths included): the two role leaders
execute disjoint exec-masked halves of P1 and P3, every live wave executes P2 (its own role's half) and P4."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1] if len(sys.argv) > 1 else "em_kernelILi0ELi1ELi2ELb0"
src = os.path.join(ROOT, "colate_amd", "csrc", "em_kernels_ilp.hip")
elf = "/tmp/isa_listing.elf"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f"-I{ROOT}/include",
                       "--offload-arch=gfx950", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-force-precise-rotation-cost=true", "--cuda-device-only", "-c",
                       "--no-gpu-bundle-output", src, "-o", elf], stderr=subprocess.DEVNULL)
dis = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", elf], text=True).split("\n")
start = [i for i, l in enumerate(dis) if re.match(r"^[0-9a-f]+ <.*" + re.escape(pat) + r".*>:$", l)][0]
end = next(i for i in range(start + 1, len(dis)) if re.match(r"^[0-9a-f]+ <.*>:$", dis[i]))
body = [l for l in dis[start + 1:end] if l.strip()]
ins = []
for l in body:
    m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", l)
    if m:
        ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
bars = [k for k, (_, op, _) in enumerate(ins) if op == "s_barrier"]
# the loop: barriers 1..3 are the last three before the epilogue's; the back edge is the last branch to an address before barrier 1
b1, b2, b3 = bars[-4], bars[-3], bars[-2]



def cls(op, args):
    if op.startswith("v_") and "dpp" in op + args:
        return "VALU dpp mov"
    if op.startswith(("v_fma_f64", "v_fmac_f64", "v_mul_f64", "v_add_f64", "v_max_f64", "v_min_f64", "v_ldexp_f64", "v_rcp_f64",
                      "v_div_", "v_rndne_f64", "v_frexp", "v_cmp_", "v_cvt_f64", "v_trig")) and "f64" in op:
        return "VALU f64"
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"):
        return "VALU lane<->scalar"
    if op.startswith("v_"):
        return "VALU other"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier")):
        return op.split("_")[1] if op != "s_nop" else "s_nop"
    if op.startswith("s_"):
        return "SALU"
    return "other"


# main loop: the backward branch behind barrier 3 whose target lies between the last prologue barrier and barrier 1
addr0 = ins[0][0]
cands = []
for k, (a, op, args) in enumerate(ins):
    if op.startswith(("s_cbranch", "s_branch")) and k > b3:
        m = re.search(r"(-?\d+)", args)
        if m:
            off = int(m.group(1))
            off = off - 65536 if off > 32767 else off  # (the disassembler prints the 16-bit field unsigned)
            t = a + 4 + 4 * off
            if ins[bars[-5]][0] < t <= ins[b1][0]:
                cands.append((t, k))
loop_start_addr = min(t for t, _ in cands)
loop_end = max(k for t, k in cands if t == loop_start_addr)
ls = next(k for k, (a, _, _) in enumerate(ins) if a >= loop_start_addr)
phases = [("P1 epoch values (role leaders; cs scan + exp_om | exp, 1/lambda, beta)  -> barrier 1", ls, b1),
          ("P2 bin terms + row-segmented reduce + tails (every live wave, own role)  -> barrier 2", b1 + 1, b2),
          ("P3 per-epoch sums, suffix scan (shared) | affine scan (not shared), partial N, D (role leaders)  -> barrier 3", b2 + 1, b3),
          ("P4 M-step, stop rule (every wave)  -> back edge", b3 + 1, loop_end)]
print(f"# EM loop of {pat} (gfx950), {ins[loop_end][0] + 4 - loop_start_addr} bytes at +0x{loop_start_addr - addr0:x} .. +0x{ins[loop_end][0] - addr0:x}")
print("# phase, static instruction counts by class (all paths: both roles' exec-masked halves and the cold paths)")
for name, lo, hi in phases:
    c = collections.Counter(cls(op, args) for _, op, args in ins[lo:hi + 1])
    tot = sum(v for k, v in c.items())
    print(f"## {name}\n#    {tot} instructions: " + ", ".join(f"{k} {v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
print()
for name, lo, hi in phases:
    print(f"\n######## {name}")
    for a, op, args in ins[lo:hi + 1]:
        print(f"  +0x{a - addr0:05x}  {op:28s} {args}")
