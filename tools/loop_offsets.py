#!/usr/bin/env python3
"""Where the EM loops sit relative to the 64-byte instruction-fetch lines, per kernel instantiation.

The time of an iteration depends on the placement of the loop's code ("code placement"): every loop compiled
per kind of wave is pinned by an anchor of its own (COLATE_LOOP_ANCHOR in em_kernel_impl.hpp: `.p2align 6` + em_loop_pad()
dwords), so an edit elsewhere cannot move it.  This tool builds the two kernel translation units as the Makefile does,
disassembles the device code and lists, for every kernel, every anchor (a run of `s_nop 0`): where the code behind it starts
relative to the 64-byte line and how far behind it the next loop head (target of a backward branch) follows.

    python tools/loop_offsets.py                    # print the table
    python tools/loop_offsets.py --write FILE.json  # ... and record it (profiles/r04_loop_offsets.json is the committed one)
    python tools/loop_offsets.py --check FILE.json  # exit 1 if a rebuild has moved a loop head relative to the recorded table
                                                    # (the pads of em_loop_pad() were tuned on exactly that placement)
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "colate_amd", "csrc")
UNITS = {"em_kernels_ilp.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"], "em_kernels.hip": []}


def target(a, args):
    off = int(re.search(r"(-?\d+)", args).group(1))
    off = off - 65536 if off > 32767 else off
    return a + 4 + 4 * off


def loops_of(unit, flags):
    elf = f"/tmp/loop_offsets_{unit}.elf"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f"-I{ROOT}/include", "--offload-arch=gfx950",
                           "-mllvm", "-force-precise-rotation-cost=true", *flags, "--cuda-device-only", "-c", "--no-gpu-bundle-output",
                           os.path.join(CSRC, unit), "-o", elf], stderr=subprocess.DEVNULL)
    dis = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", elf], text=True).split("\n")
    out = {}
    name, ins = None, []

    def flush():
        if name is None or "em_kernel" not in name:
            return
        demangled = subprocess.check_output(["c++filt", name], text=True).strip()
        short = re.sub(r".*(em_kernel<[^>]*>).*", r"\1", demangled)
        # loop heads = targets of backward branches; anchors = runs of two or more `s_nop 0` (what `.p2align 6` + the pad emit)
        heads = sorted({target(a, args) for a, op, args in ins if op.startswith(("s_cbranch", "s_branch")) and target(a, args) < a})
        found = []
        i = 0
        while i < len(ins):
            if ins[i][1] == "s_nop" and ins[i][2].strip() == "0":
                j = i
                while j < len(ins) and ins[j][1] == "s_nop" and ins[j][2].strip() == "0":
                    j += 1
                if j - i >= 2 and j < len(ins):
                    after = ins[j][0]
                    nxt = [h for h in heads if h >= after]
                    found.append({"anchor_end_mod_64": after % 64, "bytes_to_next_loop_head": (nxt[0] - after) if nxt else None,
                                  "head_mod_64": (nxt[0] % 64) if nxt else None})
                i = j
            else:
                i += 1
        out[f"{unit}: {short}"] = found

    for l in dis:
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", l)
        if m:
            flush()
            name, ins = m.group(1), []
            continue
        m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", l)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    flush()
    return out


def main():
    table = {}
    for unit, flags in UNITS.items():
        table.update(loops_of(unit, flags))
    for k, loops in table.items():
        print(f"{k}: {len(loops)} anchors; (anchor end mod 64, bytes to the next loop head): " +
              " ".join(f"({f['anchor_end_mod_64']},{f['bytes_to_next_loop_head']})" for f in loops))
    if "--write" in sys.argv:
        json.dump(table, open(sys.argv[sys.argv.index("--write") + 1], "w"), indent=1)
    if "--check" in sys.argv:
        ref = json.load(open(sys.argv[sys.argv.index("--check") + 1]))
        bad = [k for k in ref if [f["head_mod_64"] for f in ref[k]] != [f["head_mod_64"] for f in table.get(k, [])]]
        if bad:
            print("LOOP PLACEMENT CHANGED relative to the table the pads were tuned on (re-run tools/pad_sweep.sh on the GPU, then "
                  "tools/loop_offsets.py --write):\n  " + "\n  ".join(bad))
            sys.exit(1)
        print("loop heads where the tuned table has them")


if __name__ == "__main__":
    main()
