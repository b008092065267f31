"""Thin Python face of the C ABI (include/colate_amd.h).  numpy arrays for the host-pointer
entry points, torch CUDA(=HIP) tensors for the *_device ones."""
import ctypes

import numpy as np

from ._lib import ColateError, c_char_p, c_int, check, lib  # noqa: F401

FLAG_NAN, FLAG_NEG, FLAG_MAXITER, FLAG_UNRESOLVED = 1, 2, 4, 8
STATUS_MASK = 0x07  # the error-like bits of out_flags (NaN / negative / iteration cap)
DEFAULT_MAX_ITER = 100000
DEFAULT_MIN_ITER = 1000
DEFAULT_REL_TOL = 1e-7
DEFAULT_RATE_FLOOR = 5e-9
DEFAULT_INIT_RATE = 1.0 / 20000.0
MAX_EPOCHS = 1024


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data


def version():
    return lib.colate_version().decode()


def device_count():
    return lib.colate_device_count()


def unresolved_epochs(flags):
    """COLATE_UNRESOLVED_EPOCHS: how many trailing epochs of each replicate are below the resolution of the
    reference's arithmetic (include/colate_amd.h, COLATE_FLAG_UNRESOLVED)."""
    return (np.asarray(flags).astype(np.int64) & 0xFFFFFFFF) >> 8


def status_flags(flags):
    """The error-like bits of out_flags (NaN / negative / iteration cap); 0 = the replicate ran clean."""
    return np.asarray(flags) & STATUS_MASK


EM_VARIANTS = ("latency-ilp", "latency", "throughput")


def em_kernel_variant(B, E):
    """Which build of the EM kernel a batch of this shape runs on the current device."""
    return EM_VARIANTS[check(lib.colate_em_kernel_variant(int(B), int(E)))]


def em_force_variant(name=None):
    """Force the build of the EM kernel for E <= 128 ("latency-ilp", "latency", "throughput"); None = automatic."""
    check(lib.colate_em_force_variant(-1 if name is None else EM_VARIANTS.index(name)))


def age_grid():
    """coal.cpp:3126-3137."""
    g = np.zeros(256)
    n = check(lib.colate_age_grid(_p(g), 256))
    return g[:n].copy()


def epochs_from_bins(bins, age=0.0, years_per_gen=28.0):
    """coal.cpp:3551-3632.  Returns (epochs, ep_null)."""
    ep = np.zeros(MAX_EPOCHS)
    en = c_int(0)
    n = check(lib.colate_epochs_from_bins(bins.encode(), age, years_per_gen, _p(ep), MAX_EPOCHS, ctypes.byref(en)))
    return ep[:n].copy(), en.value


def epochs_from_coal(path, age=0.0):
    """coal.cpp:3508-3549, 3638-3646.  Returns (epochs, init_rates)."""
    ep = np.zeros(MAX_EPOCHS)
    r = np.zeros(MAX_EPOCHS)
    n = check(lib.colate_epochs_from_coal(str(path).encode(), age, _p(ep), _p(r), MAX_EPOCHS))
    return ep[:n].copy(), r[:n].copy()


class Rng:
    """std::mt19937 handle (coal.cpp:3157-3162)."""

    def __init__(self, seed):
        self.h = lib.colate_rng_create(seed)

    def __del__(self):
        if getattr(self, "h", None):
            lib.colate_rng_destroy(self.h)
            self.h = None


def bootstrap_counts(rng, num_bootstrap, age_grid_, age, sh_block, ns_block, sh_emp_block, ns_emp_block):
    """coal.cpp:3344-3451.  Block tables are [nb][A].  Returns (cnt_shared[B][A], cnt_notshared[B][A])."""
    g = _f64(age_grid_)
    t = [_f64(x) for x in (sh_block, ns_block, sh_emp_block, ns_emp_block)]
    nb, A = t[0].shape
    csh = np.zeros((num_bootstrap, A))
    cns = np.zeros((num_bootstrap, A))
    check(lib.colate_bootstrap_counts(rng.h, num_bootstrap, nb, A, _p(g), age, _p(t[0]), _p(t[1]), _p(t[2]),
                                      _p(t[3]), _p(csh), _p(cns)))
    return csh, cns


def bootstrap_weights(rng, num_bootstrap, nb):
    """coal.cpp:3350-3357: multinomial block weights [num_bootstrap][nb] from the shared mt19937."""
    w = np.zeros((num_bootstrap, nb))
    check(lib.colate_bootstrap_weights(rng.h, num_bootstrap, nb, _p(w)))
    return w


def bootstrap_counts_device(age_grid_, age, weights, sh_block, ns_block, sh_emp_block, ns_emp_block, cnt_shared,
                            cnt_notshared, stream=None):
    """colate_bootstrap_counts_device on torch tensors in HBM (float64, contiguous); asynchronous."""
    B, nb = weights.shape
    A = age_grid_.numel()
    for t in (age_grid_, weights, sh_block, ns_block, sh_emp_block, ns_emp_block, cnt_shared, cnt_notshared):
        assert t.is_cuda and t.is_contiguous()
    import torch

    status = torch.zeros(1, dtype=torch.int32, device=weights.device)
    check(lib.colate_bootstrap_counts_device(B, nb, A, age_grid_.data_ptr(), float(age), weights.data_ptr(),
                                             sh_block.data_ptr(), ns_block.data_ptr(), sh_emp_block.data_ptr(),
                                             ns_emp_block.data_ptr(), cnt_shared.data_ptr(), cnt_notshared.data_ptr(),
                                             status.data_ptr(), _stream_ptr(stream)))
    return status


def bootstrap_counts_from_weights(age_grid_, age, weights, sh_block, ns_block, sh_emp_block, ns_emp_block):
    """coal.cpp:3358-3451 on the host from given weights[B][nb] (the host twin of the GPU bootstrap)."""
    g, w = _f64(age_grid_), _f64(np.atleast_2d(weights))
    t = [_f64(x) for x in (sh_block, ns_block, sh_emp_block, ns_emp_block)]
    nb, A = t[0].shape
    B = w.shape[0]
    assert w.shape == (B, nb)
    csh = np.zeros((B, A))
    cns = np.zeros((B, A))
    check(lib.colate_bootstrap_counts_from_weights(B, nb, A, _p(g), float(age), _p(w), _p(t[0]), _p(t[1]), _p(t[2]), _p(t[3]),
                                                   _p(csh), _p(cns)))
    return csh, cns


def bootstrap_em_batch(age_grid_, age, weights, sh_block, ns_block, sh_emp_block, ns_emp_block, epochs, init_rates=None,
                       max_iter=DEFAULT_MAX_ITER, min_iter=DEFAULT_MIN_ITER, rel_tol=DEFAULT_REL_TOL,
                       rate_floor=DEFAULT_RATE_FLOOR, want_counts=False):
    """colate_bootstrap_em_batch: block tables [nb][A] + weights [B][nb] -> (rates, iters, loglik, flags[, csh, cns])."""
    g, w, ep = _f64(age_grid_), _f64(np.atleast_2d(weights)), _f64(epochs)
    t = [_f64(x) for x in (sh_block, ns_block, sh_emp_block, ns_emp_block)]
    nb, A = t[0].shape
    B, E = w.shape[0], ep.size
    init = _f64(np.full(E, DEFAULT_INIT_RATE) if init_rates is None else init_rates)
    rates, iters, ll, flags = np.zeros((B, E)), np.zeros(B, dtype=np.int32), np.zeros(B), np.zeros(B, dtype=np.int32)
    csh, cns = (np.zeros((B, A)), np.zeros((B, A))) if want_counts else (None, None)
    check(lib.colate_bootstrap_em_batch(B, nb, E, A, _p(g), float(age), _p(w), _p(t[0]), _p(t[1]), _p(t[2]), _p(t[3]), _p(ep),
                                        _p(init), max_iter, min_iter, rel_tol, rate_floor, _p(rates), _p(iters), _p(ll),
                                        _p(flags), _p(csh) if want_counts else None, _p(cns) if want_counts else None))
    return (rates, iters, ll, flags, csh, cns) if want_counts else (rates, iters, ll, flags)


def bootstrap_em_batch_groups(age_grid_, ages, weights, tables, epochs, init_rates=None, max_iter=DEFAULT_MAX_ITER,
                              min_iter=DEFAULT_MIN_ITER, rel_tol=DEFAULT_REL_TOL, rate_floor=DEFAULT_RATE_FLOOR,
                              want_counts=False):
    """colate_bootstrap_em_batch_groups (batched all-pairs): per group g a sample age ages[g], weights[g] = [B][nb_g],
    tables[g] = (sh, ns, sh_emp, ns_emp) each [nb_g][A], epochs[g] = [E].  Row g * B + i of the outputs = replicate i
    of group g."""
    g = _f64(age_grid_)
    G = len(weights)
    ws = [_f64(np.atleast_2d(w)) for w in weights]
    B = ws[0].shape[0]
    nb = np.ascontiguousarray([w.shape[1] for w in ws], dtype=np.int32)
    assert all(w.shape[0] == B for w in ws)
    ep = _f64(np.atleast_2d(epochs))
    E, A = ep.shape[1], g.size
    assert ep.shape == (G, E)
    init = _f64(np.full((G, E), DEFAULT_INIT_RATE) if init_rates is None else np.atleast_2d(init_rates))
    w_all = _f64(np.concatenate([w.ravel() for w in ws]))
    t_all = [_f64(np.concatenate([_f64(tables[k][j]).reshape(-1, A) for k in range(G)])) for j in range(4)]
    assert all(t.shape == (int(nb.sum()), A) for t in t_all)
    a = _f64(ages)
    R = G * B
    rates, iters, ll, flags = np.zeros((R, E)), np.zeros(R, dtype=np.int32), np.zeros(R), np.zeros(R, dtype=np.int32)
    csh, cns = (np.zeros((R, A)), np.zeros((R, A))) if want_counts else (None, None)
    check(lib.colate_bootstrap_em_batch_groups(G, B, E, A, _p(g), _p(nb), _p(a), _p(w_all), _p(t_all[0]), _p(t_all[1]),
                                               _p(t_all[2]), _p(t_all[3]), _p(ep), _p(init), max_iter, min_iter, rel_tol,
                                               rate_floor, _p(rates), _p(iters), _p(ll), _p(flags),
                                               _p(csh) if want_counts else None, _p(cns) if want_counts else None))
    return (rates, iters, ll, flags, csh, cns) if want_counts else (rates, iters, ll, flags)


def bootstrap_counts_groups_device(G, B, group_first, row_lo, row_hi, age_grid_, group_nb, group_block_off, group_weight_off, group_age,
                                   weights, sh_block, ns_block, sh_emp_block, ns_emp_block, cnt_shared, cnt_notshared, status, stream=None):
    """colate_bootstrap_counts_groups_device on torch tensors in HBM: the block bootstrap of rows [row_lo, row_hi) of G groups x B
    replicates (group arrays from group `group_first` on: int32 nb, int64 block / weight offsets, float64 ages); asynchronous."""
    A = age_grid_.numel()
    for t in (age_grid_, group_nb, group_block_off, group_weight_off, group_age, weights, sh_block, ns_block, sh_emp_block, ns_emp_block,
              cnt_shared, cnt_notshared, status):
        assert t.is_cuda and t.is_contiguous()
    check(lib.colate_bootstrap_counts_groups_device(int(G), int(B), int(group_first), int(row_lo), int(row_hi), A, age_grid_.data_ptr(),
                                                    group_nb.data_ptr(), group_block_off.data_ptr(), group_weight_off.data_ptr(),
                                                    group_age.data_ptr(), weights.data_ptr(), sh_block.data_ptr(), ns_block.data_ptr(),
                                                    sh_emp_block.data_ptr(), ns_emp_block.data_ptr(), cnt_shared.data_ptr(),
                                                    cnt_notshared.data_ptr(), status.data_ptr(), _stream_ptr(stream)))


def write_coal(path, epochs, rates, is_ancient=False, ep_null=0):
    e = _f64(epochs)
    r = _f64(np.atleast_2d(rates))
    check(lib.colate_write_coal(str(path).encode(), r.shape[0], e.size, _p(e), _p(r), int(is_ancient), ep_null))


def mut_main(argv):
    """The `Colate --mode mut ...` command line (argv without the program name)."""
    args = [b"Colate"] + [str(a).encode() for a in argv]
    arr = (c_char_p * len(args))(*args)
    return lib.colate_mut_main(len(args), arr)


def em_batch(age_grid_, cnt_shared, cnt_notshared, epochs, init_rates=None, max_iter=DEFAULT_MAX_ITER,
             min_iter=DEFAULT_MIN_ITER, rel_tol=DEFAULT_REL_TOL, rate_floor=DEFAULT_RATE_FLOOR):
    """colate_em_batch on host arrays.  Returns (rates[B][E], iters[B], loglik[B], flags[B])."""
    g, sh, ns, ep = _f64(age_grid_), _f64(np.atleast_2d(cnt_shared)), _f64(np.atleast_2d(cnt_notshared)), _f64(epochs)
    B, A = sh.shape
    E = ep.size
    init = _f64(np.full(E, DEFAULT_INIT_RATE) if init_rates is None else init_rates)
    rates = np.zeros((B, E))
    iters = np.zeros(B, dtype=np.int32)
    ll = np.zeros(B)
    flags = np.zeros(B, dtype=np.int32)
    check(lib.colate_em_batch(B, E, A, _p(g), _p(sh), _p(ns), _p(ep), _p(init), max_iter, min_iter, rel_tol,
                              rate_floor, _p(rates), _p(iters), _p(ll), _p(flags)))
    return rates, iters, ll, flags


def em_batch_rows(age_grid_, cnt_shared, cnt_notshared, epochs_rows, init_rows=None, max_iter=DEFAULT_MAX_ITER,
                  min_iter=DEFAULT_MIN_ITER, rel_tol=DEFAULT_REL_TOL, rate_floor=DEFAULT_RATE_FLOOR):
    """colate_em_batch_rows: epochs[B][E] (and starting rates) per replicate -- batched all-pairs."""
    g, sh, ns, ep = _f64(age_grid_), _f64(np.atleast_2d(cnt_shared)), _f64(np.atleast_2d(cnt_notshared)), _f64(epochs_rows)
    B, A = sh.shape
    E = ep.shape[1]
    assert ep.shape == (B, E)
    init = _f64(np.full((B, E), DEFAULT_INIT_RATE) if init_rows is None else init_rows)
    rates = np.zeros((B, E))
    iters = np.zeros(B, dtype=np.int32)
    ll = np.zeros(B)
    flags = np.zeros(B, dtype=np.int32)
    check(lib.colate_em_batch_rows(B, E, A, _p(g), _p(sh), _p(ns), _p(ep), _p(init), max_iter, min_iter, rel_tol,
                                   rate_floor, _p(rates), _p(iters), _p(ll), _p(flags)))
    return rates, iters, ll, flags


def em_batch_sharded(devices, age_grid_, cnt_shared, cnt_notshared, epochs, init_rates=None, max_iter=DEFAULT_MAX_ITER,
                     min_iter=DEFAULT_MIN_ITER, rel_tol=DEFAULT_REL_TOL, rate_floor=DEFAULT_RATE_FLOOR):
    """colate_em_batch_sharded: one process drives the GPUs listed in `devices` (ordinals may repeat)."""
    g, sh, ns, ep = _f64(age_grid_), _f64(np.atleast_2d(cnt_shared)), _f64(np.atleast_2d(cnt_notshared)), _f64(epochs)
    B, A = sh.shape
    E = ep.size
    init = _f64(np.full(E, DEFAULT_INIT_RATE) if init_rates is None else init_rates)
    dev = np.ascontiguousarray(devices, dtype=np.int32)
    rates = np.zeros((B, E))
    iters = np.zeros(B, dtype=np.int32)
    ll = np.zeros(B)
    flags = np.zeros(B, dtype=np.int32)
    check(lib.colate_em_batch_sharded(dev.size, _p(dev), B, E, A, _p(g), _p(sh), _p(ns), _p(ep), _p(init), max_iter,
                                      min_iter, rel_tol, rate_floor, _p(rates), _p(iters), _p(ll), _p(flags)))
    return rates, iters, ll, flags


def em_batch_rows_sharded(devices, age_grid_, cnt_shared, cnt_notshared, epochs, init_rates=None,
                          max_iter=DEFAULT_MAX_ITER, min_iter=DEFAULT_MIN_ITER, rel_tol=DEFAULT_REL_TOL,
                          rate_floor=DEFAULT_RATE_FLOOR):
    """colate_em_batch_rows_sharded: per-row epochs[B][E] (batched pairs), rows sharded over `devices`."""
    g, sh, ns = _f64(age_grid_), _f64(np.atleast_2d(cnt_shared)), _f64(np.atleast_2d(cnt_notshared))
    ep = _f64(np.atleast_2d(epochs))
    B, A = sh.shape
    E = ep.shape[1]
    assert ep.shape == (B, E)
    init = _f64(np.full((B, E), DEFAULT_INIT_RATE) if init_rates is None else np.atleast_2d(init_rates))
    assert init.shape == (B, E)
    dev = np.ascontiguousarray(devices, dtype=np.int32)
    rates = np.zeros((B, E))
    iters = np.zeros(B, dtype=np.int32)
    ll = np.zeros(B)
    flags = np.zeros(B, dtype=np.int32)
    check(lib.colate_em_batch_rows_sharded(dev.size, _p(dev), B, E, A, _p(g), _p(sh), _p(ns), _p(ep), _p(init), max_iter,
                                           min_iter, rel_tol, rate_floor, _p(rates), _p(iters), _p(ll), _p(flags)))
    return rates, iters, ll, flags


def em_estep(age_grid_, cnt_shared, cnt_notshared, epochs, rates):
    """colate_em_estep on host arrays: rates[B][E] -> (num[B][E], den[B][E], loglik[B], flags[B])."""
    g, sh, ns, ep = _f64(age_grid_), _f64(np.atleast_2d(cnt_shared)), _f64(np.atleast_2d(cnt_notshared)), _f64(epochs)
    r = _f64(np.atleast_2d(rates))
    B, A = sh.shape
    E = ep.size
    assert r.shape == (B, E)
    num = np.zeros((B, E))
    den = np.zeros((B, E))
    ll = np.zeros(B)
    flags = np.zeros(B, dtype=np.int32)
    check(lib.colate_em_estep(B, E, A, _p(g), _p(sh), _p(ns), _p(ep), _p(r), _p(num), _p(den), _p(ll), _p(flags)))
    return num, den, ll, flags


def _stream_ptr(stream):
    if stream is None:
        import torch

        stream = torch.cuda.current_stream()
    return ctypes.c_void_p(stream.cuda_stream)


def em_batch_device(age_grid_, cnt_shared, cnt_notshared, epochs, init_rates, out_rates, out_iters, out_loglik,
                    out_flags, max_iter=DEFAULT_MAX_ITER, min_iter=DEFAULT_MIN_ITER, rel_tol=DEFAULT_REL_TOL,
                    rate_floor=DEFAULT_RATE_FLOOR, stream=None):
    """colate_em_batch_device on torch tensors resident in HBM (float64 / int32, contiguous).
    Asynchronous on `stream` (default: torch's current stream)."""
    B, A = cnt_shared.shape
    E = epochs.shape[-1]
    for t in (age_grid_, cnt_shared, cnt_notshared, epochs, init_rates, out_rates, out_iters, out_loglik, out_flags):
        assert t.is_cuda and t.is_contiguous()
    check(lib.colate_em_batch_device(B, E, A, age_grid_.data_ptr(), cnt_shared.data_ptr(), cnt_notshared.data_ptr(),
                                     epochs.data_ptr(), int(epochs.dim() == 2), init_rates.data_ptr(),
                                     int(init_rates.dim() == 2), max_iter, min_iter, rel_tol, rate_floor,
                                     out_rates.data_ptr(), out_iters.data_ptr(), out_loglik.data_ptr(),
                                     out_flags.data_ptr(), _stream_ptr(stream)))


def em_estep_device(age_grid_, cnt_shared, cnt_notshared, epochs, rates, num_acc, den_acc, loglik, flags, stream=None):
    B, A = cnt_shared.shape
    E = epochs.shape[-1]
    for t in (age_grid_, cnt_shared, cnt_notshared, epochs, rates, num_acc, den_acc, loglik, flags):
        assert t.is_cuda and t.is_contiguous()
    check(lib.colate_em_estep_device(B, E, A, age_grid_.data_ptr(), cnt_shared.data_ptr(), cnt_notshared.data_ptr(),
                                     epochs.data_ptr(), rates.data_ptr(), num_acc.data_ptr(), den_acc.data_ptr(),
                                     loglik.data_ptr(), flags.data_ptr(), _stream_ptr(stream)))


class coal_EM:
    """Mirror of the reference's `class coal_EM` (include/coal/coal_EM.hpp:14-63) on top of the GPU
    E-step, so that parity tests read like include/test/test_aDNA.cpp: construct with (epochs, rates),
    call EM_shared / EM_notshared(age_begin, age_end, num, denom) -> log-normaliser.

    Only age_begin == age_end is implemented -- the only way mut() calls it (coal.cpp:3708, 3721).
    A call is one E-step over a one-bin age grid with count 1, so num/denom/logl are exactly the
    reference's per-bin outputs.  `EM_many` evaluates a whole list of ages in one launch."""

    def __init__(self, epochs, coal):
        self.epochs = _f64(epochs).copy()
        self.coal_rates = _f64(coal).copy()

    def UpdateCoal(self, coal):
        self.coal_rates = _f64(coal).copy()

    def EM_many(self, ages, shared):
        ages = _f64(np.atleast_1d(ages))
        n = ages.size
        # replicate i sees only age i: grid = all ages (sorted), count 1 at its own bin
        order = np.argsort(ages, kind="stable")
        grid = ages[order]
        cnt = np.zeros((n, n))
        cnt[np.arange(n), np.argsort(order, kind="stable")] = 1.0
        zero = np.zeros((n, n))
        rates = np.tile(self.coal_rates, (n, 1))
        sh, ns = (cnt, zero) if shared else (zero, cnt)
        num, den, ll, flags = em_estep(grid, sh, ns, self.epochs, rates)
        return num, den, ll, flags

    def _one(self, age_begin, age_end, num, denom, shared):
        if age_begin != age_end:
            raise NotImplementedError("colate_amd implements the age_begin == age_end path of coal_EM only "
                                      "(the one mut() uses, coal.cpp:3708/3721)")
        n, d, ll, _ = self.EM_many([age_begin], shared)
        num[:] = n[0]
        denom[:] = d[0]
        return float(ll[0])

    def EM_shared(self, age_begin, age_end, num, denom):
        return self._one(age_begin, age_end, num, denom, True)

    def EM_notshared(self, age_begin, age_end, num, denom):
        return self._one(age_begin, age_end, num, denom, False)
