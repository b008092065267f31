import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Tests marked `gpu` need a HIP device: skip them (visibly) where there is none instead of failing."""
    if not any("gpu" in item.keywords for item in items):
        return
    try:
        import colate_amd

        have = colate_amd.device_count() >= 1
    except Exception:  # noqa: BLE001  (no library / no driver: same answer)
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="no HIP device in this environment (run with -m gpu on the MI355X box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def _ensure_built():
    """The C-ABI library and the oracle are built in-tree; build them if a checkout is fresh."""
    lib = os.path.join(ROOT, "colate_amd", "lib", "libcolate_amd.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "colate_amd", "csrc")], stdout=subprocess.DEVNULL)
    ora = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(ora):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], stdout=subprocess.DEVNULL)


_ensure_built()
