"""The steady-state loops of the EM kernels sit where the pads of em_loop_pad() were tuned: tools/loop_offsets.py
rebuilds the two kernel translation units as the Makefile does, disassembles them and compares every loop head's offset inside its
64-byte fetch line with the table recorded on the tuned build (profiles/r04_loop_offsets.json).  A kernel edit that moves a loop makes
this fail until the pads have been swept again on the GPU (tools/pad_sweep.sh) and the table re-recorded (--write)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc (cross-compiles without a GPU)")
def test_loop_heads_are_where_the_pads_were_tuned():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "loop_offsets.py"), "--check", os.path.join(ROOT, "profiles", "r04_loop_offsets.json")],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "loop heads where the tuned table has them" in r.stdout
