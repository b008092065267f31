"""ctypes binding of libcolate_amd.so (include/colate_amd.h).

The library is built in-tree by colate_amd/csrc/Makefile (or __graft_entry__.build()).
There is no fallback: if the shared object is missing, importing this module raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# COLATE_AMD_LIB=<path>: load another build of the same library (tools/ab_bench.sh compares builds this way
# instead of copying candidates over the product library)
LIB_PATH = os.environ.get("COLATE_AMD_LIB") or os.path.join(_HERE, "lib", "libcolate_amd.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `make -C colate_amd/csrc` "
        "(or `python -c 'import __graft_entry__ as g; g.build()'`). colate_amd has no CPU fallback."
    )

# One HIP runtime per process.  A PyTorch-ROCm wheel carries its own libamdhip64 / libhsa-runtime64; libcolate_amd.so is linked
# against the system's.  Whichever is loaded first serves both (same sonames) -- unless it is the system's: torch then still
# loads its bundled copies by path, a second HSA runtime comes up in the process, finds the device taken and reports "No HIP GPUs
# are available" (tools/study/hip_runtime_order.py).  So where torch is installed it goes first; nothing else of it is used here.
try:
    import torch  # noqa: F401
except Exception:  # noqa: BLE001  (no torch: the system runtime is the only one)
    pass

lib = ctypes.CDLL(LIB_PATH)

c_int = ctypes.c_int
c_double = ctypes.c_double
c_void_p = ctypes.c_void_p
c_char_p = ctypes.c_char_p
dp = ctypes.POINTER(ctypes.c_double)
ip = ctypes.POINTER(ctypes.c_int)

# every symbol include/colate_amd.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "colate_version": (c_char_p, []),
    "colate_last_error": (c_char_p, []),
    "colate_device_count": (c_int, []),
    "colate_set_device": (c_int, [c_int]),
    "colate_warm_up": (c_int, [c_int]),
    "colate_device_touched": (c_int, []),
    "colate_em_kernel_variant": (c_int, [c_int, c_int]),
    "colate_em_force_variant": (c_int, [c_int]),
    "colate_em_batch": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_int, c_int, c_double, c_double, c_void_p, c_void_p, c_void_p, c_void_p]),
    "colate_em_batch_device": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                       c_void_p, c_int, c_int, c_int, c_double, c_double, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p]),
    "colate_em_batch_rows": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_int, c_int, c_double, c_double, c_void_p, c_void_p, c_void_p, c_void_p]),
    "colate_em_batch_sharded": (c_int, [c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_int, c_int, c_double, c_double, c_void_p, c_void_p, c_void_p, c_void_p]),
    "colate_em_batch_rows_sharded": (c_int, [c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_int, c_int, c_double, c_double, c_void_p, c_void_p, c_void_p,
                                             c_void_p]),
    "colate_em_estep": (c_int, [c_int, c_int, c_int] + [c_void_p] * 9),
    "colate_em_estep_device": (c_int, [c_int, c_int, c_int] + [c_void_p] * 10),
    "colate_age_grid": (c_int, [c_void_p, c_int]),
    "colate_epochs_from_bins": (c_int, [c_char_p, c_double, c_double, c_void_p, c_int, ip]),
    "colate_epochs_from_coal": (c_int, [c_char_p, c_double, c_void_p, c_void_p, c_int]),
    "colate_rng_create": (c_void_p, [ctypes.c_uint]),
    "colate_rng_destroy": (None, [c_void_p]),
    "colate_bootstrap_counts": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_double] + [c_void_p] * 6),
    "colate_bootstrap_weights": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "colate_bootstrap_counts_device": (c_int, [c_int, c_int, c_int, c_void_p, c_double] + [c_void_p] * 9),
    "colate_bootstrap_em_batch": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_double] + [c_void_p] * 7
                                  + [c_int, c_int, c_double, c_double] + [c_void_p] * 6),
    "colate_bootstrap_counts_from_weights": (c_int, [c_int, c_int, c_int, c_void_p, c_double] + [c_void_p] * 7),
    "colate_bootstrap_em_batch_groups": (c_int, [c_int, c_int, c_int, c_int] + [c_void_p] * 10
                                         + [c_int, c_int, c_double, c_double] + [c_void_p] * 6),
    "colate_bootstrap_counts_groups_device": (c_int, [c_int] * 6 + [c_void_p] * 14),
    "colate_bootstrap_em_batch_groups_allgather": (c_int, [c_void_p] + [c_int] * 6 + [c_void_p] * 10
                                                   + [c_int, c_int, c_double, c_double] + [c_void_p] * 4),
    "colate_release_workspace": (c_int, []),
    "colate_shard_bounds": (c_int, [c_int, c_int, c_int, ip, ip]),
    "colate_comm_unique_id": (c_int, [c_void_p]),
    "colate_comm_create": (c_int, [c_void_p, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "colate_comm_destroy": (c_int, [c_void_p]),
    "colate_em_batch_allgather": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_int, c_int, c_double, c_double, c_void_p, c_void_p, c_void_p, c_void_p]),
    "colate_bootstrap_em_batch_allgather": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_double] + [c_void_p] * 7
                                            + [c_int, c_int, c_double, c_double] + [c_void_p] * 4),
    "colate_write_coal": (c_int, [c_char_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int]),
    "colate_mut_main": (c_int, [c_int, ctypes.POINTER(c_char_p)]),
}

for _name, (_res, _args) in SIGNATURES.items():
    try:
        _f = getattr(lib, _name)  # AttributeError here = header and library out of sync
    except AttributeError:
        if os.environ.get("COLATE_AMD_LIB"):  # an older build under comparison may lack newer entry points
            continue
        raise
    _f.restype = _res
    _f.argtypes = _args


class ColateError(RuntimeError):
    def __init__(self, code):
        self.code = code
        msg = lib.colate_last_error()
        super().__init__(f"colate_amd error {code}: {msg.decode() if msg else ''}")


def check(rc):
    if rc < 0:
        raise ColateError(rc)
    return rc
