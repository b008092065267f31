#!/usr/bin/env python3
"""ISA of the steady-state EM loops of one kernel instantiation -- one loop per kind of wave (role x leadership x keeps the
verdict history), see the end of em_kernel in colate_amd/csrc/em_kernel_impl.hpp -- cut into the phases of an iteration at
the s_barrier instructions, with instruction-class counts per phase and every branch marked with its direction: the listing
committed as profiles/r02_isa_loop_<name>.txt.

    python tools/isa_listing.py 'em_kernelILi0ELi1ELi2ELb0' > profiles/r02_isa_loop_e23_latency_ilp.txt
    PICK=max NBAR=2 EXTRA_FLAGS=-DCOLATE_NO_LL_LOOPS python tools/isa_listing.py 'em_kernelILi0ELi1ELi2ELb0ELi2E' > profiles/r03_isa_loop_e23_latency_ilp.txt
(round 3: the build for batches that leave every workgroup a CU has two barriers per steady-state iteration and three in the
loops of the log-likelihood phase, whose out-of-line blocks confuse the "shorter of two loops" rule below: list it with the
log-likelihood loops compiled out -- same steady-state loops, other addresses)

Builds colate_amd/csrc/em_kernels_ilp.hip with the Makefile's flags (+ line tables) to a device ELF and disassembles it
(llvm-objdump -l).  A loop is a backward branch over exactly three barriers; the COLATE_BOTH(...) line its iteration counter
is attributed to tells which kind of wave it is for; its address range is listed whole, so a block the compiler placed out of line but inside the range
(role B's leader has some) shows up between the phases: blocks are separated by a blank line wherever a branch target or
the instruction behind an unconditional branch starts one."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1] if len(sys.argv) > 1 else "em_kernelILi0ELi1ELi2ELb0"
src = os.path.join(ROOT, "colate_amd", "csrc", os.environ.get("SRC", "em_kernels_ilp.hip"))  # SRC=em_kernels.hip NOILP=1: the default-scheduler builds (throughput variant)
impl = os.path.join(ROOT, "colate_amd", "csrc", "em_kernel_impl.hpp")
elf = "/tmp/isa_listing.elf"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f"-I{ROOT}/include",
                       "--offload-arch=gfx950", *([] if os.environ.get("NOILP") else ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]), "-mllvm", "-force-precise-rotation-cost=true",
                       "-gline-tables-only", *os.environ.get("EXTRA_FLAGS", "").split(), "--cuda-device-only", "-c", "--no-gpu-bundle-output", src, "-o", elf], stderr=subprocess.DEVNULL)
dis = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "-l", elf], text=True).split("\n")
start = [i for i, l in enumerate(dis) if re.match(r"^[0-9a-f]+ <.*" + re.escape(pat) + r".*>:$", l)][0]
end = next(i for i in range(start + 1, len(dis)) if re.match(r"^[0-9a-f]+ <.*>:$", dis[i]))
impl_lines = open(impl).read().split("\n")
ins = []  # (address, op, args, source line in em_kernel_impl.hpp or None)
cur = None
for l in dis[start + 1:end]:
    m = re.match(r"^; (/\S+):(\d+)", l)
    if m:
        cur = int(m.group(2)) if m.group(1).endswith("em_kernel_impl.hpp") else None
        continue
    m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", l)
    if m:
        ins.append((int(m.group(3), 16), m.group(1), m.group(2), cur))
addr0 = ins[0][0]
index = {a: k for k, (a, _, _, _) in enumerate(ins)}


def target(a, args):
    off = int(re.search(r"(-?\d+)", args).group(1))
    off = off - 65536 if off > 32767 else off  # (the disassembler prints the 16-bit field unsigned)
    return a + 4 + 4 * off


def cls(op, args):
    if op.startswith("v_") and "dpp" in op + args:
        return "VALU dpp mov"
    if op.startswith("v_") and "f64" in op:
        return "VALU f64"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "VALU lane<->scalar"
    if op.startswith("v_"):
        return "VALU other"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op in ("s_waitcnt", "s_nop", "s_barrier"):
        return op
    if op.startswith("s_"):
        return "SALU"
    return "other"


kinds = {"C0, C2, C1, C1": "role A (shared), split: even epochs, keeps the verdict history (wave 0)", "C1, C2, C0, C2": "role B (not shared), split: even epochs (wave 1)",
         "C0, C3, C0, C1": "role A, split: odd epochs (wave 2)", "C1, C3, C0, C2": "role B, split: odd epochs (wave 3)",
         "C0, C1, C0, C1": "role A (shared) leader (wave 0)", "C1, C1, C1, C0": "role B (not shared) leader whose epoch values wave 3 computes, keeps the verdict history (wave 1)",
         "C1, C0, C0, C2": "role B second bin group, computes role B's epoch values (wave 3)",
         "C1, C1, C0, C2": "role B (not shared) leader that computes its epoch values",
         "C0, C1, C1, C1": "role A leader that also keeps the verdict history (one bin group only)",
         "C0, C0, C0, C1": "role A second and further bin groups, compute role A's epoch values for themselves (wave 2)", "C1, C0, C0, C0": "role B further bin groups",
         "C0, C0, C1, C0": "role A second bin group, keeps the verdict history", "C0, C0, C0, C0": "role A further bin groups"}
print(f"# Steady-state EM loops of {pat} (gfx950), built as the Makefile builds em_kernels_ilp.hip")
# every backward branch whose range holds exactly the three barriers of an iteration is a loop; the COLATE_BOTH(...) line
# that some instruction of it (the iteration counter) is attributed to tells the kind of wave; of the two loops of a kind
# the shorter one is the steady-state loop (no log-likelihood)
found = collections.defaultdict(list)
for k, (a, op, args, line) in enumerate(ins):
    if not op.startswith(("s_cbranch", "s_branch")):
        continue
    t = target(a, args)
    if t > a or t not in index:
        continue
    k0 = index[t]
    nbar = sum(1 for x in ins[k0:k] if x[1] == "s_barrier")
    if nbar != int(os.environ.get("NBAR", "3")):  # (NBAR=2: the build for B <= #CUs at up to 64 epochs has no barrier between the epoch values and the bin terms)
        continue
    tags = {m.group(1) for x in ins[k0:k + 1] if x[3] for m in [re.search(r"COLATE_BOTH(?:_B)?\(([^)]*)\)", impl_lines[x[3] - 1])] if m}
    if len(tags) == 1:
        found[tags.pop()].append((nbar, k - k0, k0, k))
loops = []
for tag, what in kinds.items():
    if found.get(tag):
        # (PICK=max: with the log-likelihood loops compiled out the only other "loops" of a kind are back edges from its out-of-line
        # blocks into the middle of the steady-state loop: shorter ranges of the same code)
        # ... and loops around it (the refresh schedule of role B's leader) are much longer: the longest range below 1.6 x the shortest)
        if os.environ.get("PICK") == "max":
            shortest = min(f[1] for f in found[tag])
            _, _, k0, k1 = max(f for f in found[tag] if f[1] < 1.6 * shortest)
        else:
            _, _, k0, k1 = min(found[tag])
        loops.append((tag, what, k0, k1))
for tag, what, k0, k1 in loops:
    seg = ins[k0:k1 + 1]
    c = collections.Counter(cls(op, args) for _, op, args, _ in seg)
    nb = sum(1 for _, op, _, _ in seg if op.startswith(("s_cbranch", "s_branch")))
    print(f"## {what}  [COLATE_BOTH({tag})]: {len(seg)} instructions, {seg[-1][0] + 4 - seg[0][0]} bytes at +0x{seg[0][0] - addr0:x}: "
          + ", ".join(f"{k} {v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
names = ["P1 epoch values (leaders: cs scan + exp_om | exp, 1/lambda, beta) + rate prefetch  -> barrier 1",
         "P2 bin terms + row-segmented reduce + tails  -> barrier 2",
         "P3 per-epoch sums, suffix scan | affine scan, partial N, D (leaders)  -> barrier 3",
         "P4 M-step  -> back edge"]
if os.environ.get("NBAR", "3") == "2":
    names = ["P1 + P2: epoch values (every role-A wave for itself; wave 3 role B's), rate / S_k fetch by ds_bpermute, bin terms + row-segmented reduce + tails -- no barrier between  -> barrier 2"] + names[2:]
for tag, what, k0, k1 in loops[:int(os.environ.get("FULL", "4"))]:  # the first four kinds in full
    print(f"\n\n################ {what}  [COLATE_BOTH({tag})]")
    seg = ins[k0:k1 + 1]
    targets = {target(a, args) for a, op, args, _ in ins if op.startswith(("s_cbranch", "s_branch"))}
    ph = 0
    print(f"\n######## {names[0]}")
    prev_uncond = False
    for a, op, args, line in seg:
        if a in targets or prev_uncond:
            print()
        note = ""
        if op.startswith(("s_cbranch", "s_branch")):
            t = target(a, args)
            inside = seg[0][0] <= t <= seg[-1][0]
            note = f"   ; -> +0x{t - addr0:05x} ({'backward' if t <= a else 'forward'}{'' if inside else ', out of the loop range'})"
        print(f"  +0x{a - addr0:05x}  {op:28s} {args}{note}")
        prev_uncond = op == "s_branch"
        if op == "s_barrier":
            ph += 1
            print(f"\n######## {names[ph]}")
