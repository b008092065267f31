#!/bin/bash
# Compiles one .hip translation unit the way hipcc does -- device code to gfx950 assembly, assembly to a code object, code
# object bundled and embedded into the host object -- with ONE step in between: align_isa.py lays the instructions out for
# the instruction fetch of a lone wave (8-byte instructions on 8-byte boundaries, loop heads on 32-byte boundaries; see
# there).  Same flags as a plain `hipcc -c`; the host half is compiled by hipcc itself.
#   build_aligned.sh <src.hip> <out.o> <hipcc flags...>
set -euo pipefail
src="$1"; out="$2"; shift 2
B=/opt/rocm/lib/llvm/bin
tmp="$(mktemp -d "${TMPDIR:-/tmp}/colate_aligned.XXXXXX")"
trap 'rm -rf "$tmp"' EXIT
/opt/rocm/bin/hipcc "$@" --offload-arch=gfx950 --cuda-device-only -S "$src" -o "$tmp/dev.s" 2> "$tmp/dev.log" || { cat "$tmp/dev.log" >&2; exit 1; }
grep -v "argument unused during compilation" "$tmp/dev.log" >&2 || true
$B/llvm-mc -triple=amdgcn-amd-amdhsa -mcpu=gfx950 -show-encoding "$tmp/dev.s" -o "$tmp/dev.enc.s"
python3 "$(dirname "$0")/align_isa.py" "$tmp/dev.enc.s" "$tmp/dev.aligned.s"
$B/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$tmp/dev.aligned.s" -o "$tmp/dev.o"
$B/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared "$tmp/dev.o" -o "$tmp/dev.co"
$B/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
  -input=/dev/null -input="$tmp/dev.co" -output="$tmp/dev.hipfb"
/opt/rocm/bin/hipcc "$@" --offload-arch=gfx950 --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang "$tmp/dev.hipfb" -c "$src" -o "$out"
if [ -n "${COLATE_KEEP_ISA:-}" ]; then cp "$tmp/dev.co" "$COLATE_KEEP_ISA"; fi
