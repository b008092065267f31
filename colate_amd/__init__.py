"""colate_amd -- MI355X-native EM path of `Colate --mode mut`.

Python here is plumbing only (ctypes over the C ABI in include/colate_amd.h, torch for
device memory / streams / torch.distributed): the product is colate_amd/csrc (HIP + C++).
"""
from ._lib import LIB_PATH, ColateError  # noqa: F401
from .api import (  # noqa: F401
    DEFAULT_INIT_RATE,
    DEFAULT_MAX_ITER,
    DEFAULT_MIN_ITER,
    DEFAULT_RATE_FLOOR,
    DEFAULT_REL_TOL,
    FLAG_MAXITER,
    FLAG_NAN,
    FLAG_NEG,
    FLAG_UNRESOLVED,
    STATUS_MASK,
    em_force_variant,
    em_kernel_variant,
    status_flags,
    unresolved_epochs,
    Rng,
    age_grid,
    bootstrap_counts,
    bootstrap_counts_device,
    bootstrap_counts_from_weights,
    bootstrap_em_batch,
    bootstrap_em_batch_groups,
    bootstrap_weights,
    coal_EM,
    device_count,
    em_batch,
    em_batch_device,
    em_batch_rows,
    em_batch_rows_sharded,
    em_batch_sharded,
    em_estep,
    em_estep_device,
    epochs_from_bins,
    epochs_from_coal,
    mut_main,
    version,
    write_coal,
)
