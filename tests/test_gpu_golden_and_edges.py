"""GPU: the HIP path against the reference's own outputs (golden fixtures), the full drop-in CLI,
edge cases and size-independent properties at BASELINE sizes.  All calls go through the C ABI."""
import os
import subprocess

import numpy as np
import pytest

import golden_lib as gl
import oracle_lib as ol

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "colate_amd", "bin", "Colate")
RATE_RTOL = 1e-6  # north_star


def _rel(a, b):
    m = np.maximum(np.abs(a), np.abs(b))
    m[m == 0] = 1.0
    return np.abs(a - b) / m


@pytest.fixture(scope="module")
def ca():
    import colate_amd

    assert colate_amd.device_count() >= 1
    return colate_amd


def test_l1_golden_estep_through_coal_EM_mirror(ca):
    """Golden vectors of coal_EM::EM_shared/EM_notshared recorded from the reference build."""
    groups = {}
    for c in gl.l1_cases():
        key = (c["epochs"].tobytes(), c["rates"].tobytes(), c["kind"])
        groups.setdefault(key, []).append(c)
    checked = 0
    for cases in groups.values():
        ep, rates, kind = cases[0]["epochs"], cases[0]["rates"], cases[0]["kind"]
        if kind == 1 and rates[-1] <= 0:
            continue
        ages = np.array([c["age"] for c in cases])
        num, den, ll, flags = ca.coal_EM(ep, rates).EM_many(ages, kind == 0)
        dt = np.append(np.diff(ep), 0.0)
        for i, c in enumerate(cases):
            if np.isnan(c["num"]).any() or np.isnan(c["denom"]).any():
                continue  # the reference asserts (aborts) on these
            assert abs(ll[i] - c["logl"]) <= 1e-9 * max(1.0, abs(c["logl"]))
            assert (np.abs(num[i] - c["num"]) <= 1e-7 * np.abs(c["num"]) + 1e-300).all()
            assert (np.abs(den[i] - c["denom"]) <= 1e-6 * np.abs(c["denom"]) + 1e-13 * dt + 1e-300).all()
            checked += 1
    assert checked > 200


@pytest.mark.parametrize("name", gl.l2_names())
def test_l2_golden_coal_text_and_iterations(ca, name):
    """Same count tables as the reference run: identical 'Total iterations' and identical .coal text
    (6 significant digits) wherever the reference's own digits are reproducible (stable_mask)."""
    c = gl.l2_case(name)
    grid = ol.age_grid()
    ep, _ = ol.epochs_from_bins(c["bins"])
    rates, iters, ll, flags = ca.em_batch(grid, c["csh"], c["cns"], ep)
    assert (ca.status_flags(flags) == 0).all()
    assert iters.tolist() == c["iterations"]
    ref_rows = [np.array(line.split()[2:], dtype=np.float64) for line in c["coal"].split("\n")[2:] if line]
    mine_rows = [np.array(("".join("%g " % x for x in r)).split(), dtype=np.float64) for r in rates]
    r0, _, _, _ = ol.em_batch(grid, c["csh"], c["cns"], ep)
    mask = ol.stable_mask(grid, c["csh"], c["cns"], ep, r0)
    unres, unstable = ol.check_rates(rates, flags, r0, mask, RATE_RTOL)
    if ep.size < 64:  # --bins 3,7,0.2: every epoch is pinned, nothing is flagged, the whole text is identical
        assert mask.all() and (flags == 0).all()
    else:  # 122 epochs: the reference's last 16-17 are rounding residue; the kernel says so itself
        assert (unstable <= 18).all() and (unres >= unstable - 1).all() and (unres <= unstable + 2).all(), (unres, unstable)
    for b in range(len(ref_rows)):
        keep = ep.size - int(unres[b])  # ... and every token the kernel does not flag is the reference's token
        assert np.array_equal(mine_rows[b][:keep], ref_rows[b][:keep])
        assert np.array_equal(mine_rows[b][mask[b]], ref_rows[b][mask[b]])


@pytest.mark.parametrize("name", gl.l2_names())
def test_colate_mat_hook_three_way(ca, name, tmp_path):
    """The reference's hook for precomputed count tables (coal.cpp:3169-3170, 3471-3499): OUR command line loads the
    same OUT.colate_mat the reference binary was given when the golden was made, so reference .coal, our .coal and the
    oracle's rates are a three-way comparison on identical counts (23 and 122 epochs)."""
    c = gl.l2_case(name)
    B = len(c["iterations"])
    grid = ol.age_grid()
    gl.write_colate_mat(tmp_path / "OUT.colate_mat", grid, c["csh"], c["cns"])
    r = subprocess.run([CLI, "--mode", "mut", "--mut", "dummy", "--bins", c["bins"], "--num_bootstraps", str(B), "-o", "OUT"],
                       cwd=str(tmp_path), capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-800:]
    err = r.stderr.decode()
    assert "Loading precomputed file OUT.colate_mat" in err
    assert [int(l.rsplit(" ", 1)[1]) for l in err.split("\n") if l.startswith("Bootstrap ")] == c["iterations"]
    mine = (tmp_path / "OUT.coal").read_text().split("\n")
    ref = c["coal"].split("\n")
    assert mine[:2] == ref[:2] and len(mine) == len(ref)
    ep, _ = ol.epochs_from_bins(c["bins"])
    r0, _, _, _ = ol.em_batch(grid, c["csh"], c["cns"], ep)
    assert gl.coal_text(ep, r0) == c["coal"]  # oracle == reference, whole text
    mask = ol.stable_mask(grid, c["csh"], c["cns"], ep, r0)
    note = [l for l in err.split("\n") if l.startswith("Note: the last ")]
    k_cli = int(note[0].split()[3]) if note else 0
    unstable = ep.size - mask.sum(axis=1)
    assert unstable.max() - 1 <= k_cli <= unstable.max() + 2
    if ep.size < 64:
        assert k_cli == 0 and mine == ref  # 23 epochs: identical files
    for b in range(B):
        m_tok, r_tok = mine[2 + b].split(), ref[2 + b].split()
        assert m_tok[: 2 + ep.size - k_cli] == r_tok[: 2 + ep.size - k_cli]


def test_colate_mat_written_by_us_read_by_both(ca, tmp_path):
    """The 6-digit, /1e3 .colate_mat of --write_colate_mat (tests/golden/l4_colate_mat: written by our CLI, read by the
    reference binary when the fixture was made): our CLI loads the same file and prints the reference's .coal."""
    import json
    import shutil

    src = os.path.join(gl.HERE, "l4_colate_mat")
    case = json.load(open(os.path.join(src, "case.json")))
    shutil.copy(os.path.join(src, "OUT.colate_mat"), str(tmp_path / "OUT.colate_mat"))
    r = subprocess.run([CLI] + case["reader_args"], cwd=str(tmp_path), capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-800:]
    err = r.stderr.decode()
    assert [int(l.rsplit(" ", 1)[1]) for l in err.split("\n") if l.startswith("Bootstrap ")] == case["iterations"]
    mine = (tmp_path / "OUT.coal").read_text().split("\n")
    ref = open(os.path.join(src, "expected.coal")).read().split("\n")
    note = [l for l in err.split("\n") if l.startswith("Note: the last ")]
    k_cli = int(note[0].split()[3]) if note else 0
    assert mine[:2] == ref[:2] and len(mine) == len(ref) and k_cli <= 4
    for m, t in zip(mine[2:], ref[2:]):
        assert m.split()[: 2 + 23 - k_cli] == t.split()[: 2 + 23 - k_cli]


@pytest.mark.parametrize("name", gl.l3_names())
def test_l3_cli_drop_in(ca, name, tmp_path):
    """`Colate --mode mut` of colate_amd on the reference's input files and --seed: same stderr
    iteration counts, and the same .coal text -- header, epochs and every rate the reference itself
    determines (its far-tail epochs are rounding residue, see oracle_lib.stable_mask / DESIGN.md §6)."""
    case = gl.l3_stage(name, str(tmp_path))
    args = list(case["args"])
    args[args.index("-o") + 1] = "mine"
    B = int(args[args.index("--num_bootstraps") + 1])
    r = subprocess.run([CLI] + args + ["--counts_out", "mine.counts"], cwd=str(tmp_path), capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-800:]
    err = r.stderr.decode()
    got_iters = [int(l.rsplit(" ", 1)[1]) for l in err.split("\n") if l.startswith("Bootstrap ")]
    assert got_iters == case["iterations"]
    mine = (tmp_path / "mine.coal").read_text().split("\n")
    ref = (tmp_path / "expected.coal").read_text().split("\n")
    grid, csh, cns = gl.read_counts(tmp_path / "mine.counts", B)
    age = 0.0
    if "--target_age" in args:
        age = float(np.float32(args[args.index("--target_age") + 1])) / 28.0
    kw = {}
    if "--coal" in args:  # warm start from a .coal file (coal.cpp:3508-3549, 3638-3646)
        ep, kw["init"] = ol.epochs_from_coal(tmp_path / args[args.index("--coal") + 1], age)
        ep_null = 0
    else:
        ep, ep_null = ol.epochs_from_bins(args[args.index("--bins") + 1], age, 28.0)
    note = [l for l in err.split("\n") if l.startswith("Note: the last ")]
    k_cli = int(note[0].split()[3]) if note else 0
    _assert_coal_is_the_references(mine, ref, grid, csh, cns, ep, ep_null, age, k_cli, kw)


def _assert_coal_is_the_references(mine, ref, grid, csh, cns, ep, ep_null, age, k_cli, kw={}, min_stable=0.88):
    """`mine` / `ref`: the lines of our .coal and of the reference's for the same inputs; `k_cli` = the number of trailing
    epochs the CLI's note declares unresolved.  With these small inputs the second-to-last epoch of some replicates has
    (almost) no data: the reference's own rate there moves by ~1 % under libm noise (stable_mask).  Everything the checker
    finds pinned must be the reference's token; the CLI's note must cover at least the epochs the checker finds unstable."""
    assert mine[:2] == ref[:2] and len(mine) == len(ref)
    B = csh.shape[0]
    r0, _, _, _ = ol.em_batch(grid, csh, cns, ep, **kw)
    mask = ol.stable_mask(grid, csh, cns, ep, r0, **kw)
    first = ep_null if age > 0 else 0  # ancient samples print epochs from ep_null on (coal.cpp:3837)
    unstable = ep.size - mask.sum(axis=1)
    assert mask.mean() > min_stable, mask.mean()  # (measured: 0.895 for l3_coal_modern, 0.942 for l3_modern, 1.0 for the others)
    assert unstable.max() - 1 <= k_cli <= unstable.max() + 3, (k_cli, unstable)
    for b in range(B):
        m_tok, r_tok = mine[2 + b].split(), ref[2 + b].split()
        assert m_tok[:2] == r_tok[:2] and len(m_tok) == len(r_tok)
        for j, e in enumerate(range(first, ep.size)):
            if m_tok[2 + j] != r_tok[2 + j]:
                assert not mask[b, e] and e >= ep.size - k_cli, (b, e, m_tok[2 + j], r_tok[2 + j])


def test_l3_pairs_drop_in(ca, tmp_path):
    """`Colate --pairs` (batched all-pairs, BASELINE configs[4]) against the REFERENCE run once per pair (fixture l3_pairs):
    every pair's iteration counts and .coal tokens are the reference's -- all pairs filled together from files read once,
    bootstrapped in one launch and fitted in one launch per epoch count (two here: a 500-year-old sample adds an epoch)."""
    meta = gl.l3_pairs_stage(str(tmp_path))
    common = ["--mode", "mut", "--mut", "P"] + meta["common_args"]
    B = int(common[common.index("--num_bootstraps") + 1])
    r = subprocess.run([CLI] + common + ["--pairs", "pairs.txt", "--counts_out", "x"], cwd=str(tmp_path), capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-800:]
    err = r.stderr.decode().split("\n")
    for k, p in enumerate(meta["pairs"]):
        got = [int(l.rsplit(" ", 1)[1]) for l in err if l.startswith(f"Pair {k + 1} Bootstrap ")]
        assert got == p["iterations"], (p["output"], got)
        mine = (tmp_path / (p["output"] + ".coal")).read_text().split("\n")
        ref = (tmp_path / f"expected_{p['output']}.coal").read_text().split("\n")
        grid, csh, cns = gl.read_counts(tmp_path / (p["output"] + ".counts"), B)
        age = max(float(np.float32(p["target_age"])), float(np.float32(p["reference_age"]))) / 28.0
        ep, ep_null = ol.epochs_from_bins(common[common.index("--bins") + 1], age, 28.0)
        note = [l for l in err if l.startswith(f"Note: pair {k + 1}: the last ")]
        k_cli = int(note[0].split()[5]) if note else 0
        _assert_coal_is_the_references(mine, ref, grid, csh, cns, ep, ep_null, age, k_cli, min_stable=0.8)


def test_edge_cases(ca):
    grid = ol.age_grid()
    ep, _ = ol.epochs_from_bins("3,7,0.2")
    A, E = grid.size, ep.size
    # B = 0 is a no-op
    r, it, ll, fl = ca.em_batch(grid, np.zeros((0, A)), np.zeros((0, A)), ep)
    assert r.shape == (0, E)
    # empty count tables: every num is 0 -> all rates copy epoch 0 (= 0); the stop rule never fires
    # (ll/prev = 0/0), so the cap ends the run, flagged -- exactly the oracle's behaviour
    r, it, ll, fl = ca.em_batch(grid, np.zeros((2, A)), np.zeros((2, A)), ep, max_iter=1200)
    r0, it0, ll0, fl0 = ol.em_batch(grid, np.zeros((2, A)), np.zeros((2, A)), ep, max_iter=1200)
    assert np.array_equal(r, r0) and (it == it0).all() and (fl & 4).all() and (fl0 & 4).all()
    # ragged: a single not-shared bin / a single shared bin / data only beyond the last epoch start
    for kind, b in ((1, 100), (0, 100), (1, 180), (0, 60)):
        csh, cns = np.zeros((1, A)), np.zeros((1, A))
        (csh if kind == 0 else cns)[0, b] = 37.5
        r, it, ll, fl = ca.em_batch(grid, csh, cns, ep, max_iter=1500)
        r0, it0, ll0, fl0 = ol.em_batch(grid, csh, cns, ep, max_iter=1500)
        assert (it == it0).all()
        m = ol.stable_mask(grid, csh, cns, ep, r0, max_iter=1500)
        assert _rel(r, r0)[m].max() < RATE_RTOL
    # minimal and maximal epoch counts, short age grids
    from colate_amd import workloads

    csh, cns = workloads.bootstrap_tables(grid, 2, nb=9, scale=1.0)
    for e_arr in (np.array([0.0, 1e3]), np.concatenate([[0.0], np.geomspace(30, 3e5, 254), [4e6]])):
        r, it, ll, fl = ca.em_batch(grid, csh, cns, e_arr, max_iter=1100)
        r0, it0, ll0, fl0 = ol.em_batch(grid, csh, cns, e_arr, max_iter=1100)
        assert (it == it0).all() and np.allclose(ll, ll0, rtol=1e-11, atol=0)
        m = ol.stable_mask(grid, csh, cns, e_arr, r0, max_iter=1100)
        assert _rel(r, r0)[m].max() < RATE_RTOL
    sub = slice(40, 151)
    r, it, ll, fl = ca.em_batch(grid[sub], csh[:, sub], cns[:, sub], ep)
    r0, it0, ll0, fl0 = ol.em_batch(grid[sub], csh[:, sub], cns[:, sub], ep)
    assert (it == it0).all() and _rel(r, r0).max() < 1e-8
    # warm start (--coal): starting rates given per epoch
    init = np.exp(np.random.default_rng(3).uniform(np.log(1e-6), np.log(1e-3), E))
    r, it, ll, fl = ca.em_batch(grid, csh, cns, ep, init_rates=init)
    r0, it0, ll0, fl0 = ol.em_batch(grid, csh, cns, ep, init=init)
    m = ol.stable_mask(grid, csh, cns, ep, r0, init=init)
    assert (it == it0).all() and _rel(r, r0)[m].max() < RATE_RTOL


@pytest.mark.parametrize("bins", ["3,7,0.2", "2,7.95,0.05"])
def test_iteration_limits_around_the_loop_hand_overs(ca, bins):
    """The kernel runs the iterations before min_iter, those from min_iter on and (for some waves) all of them in
    different loops (em_kernel_impl.hpp, COLATE_BOTH): every combination of the two limits that moves the hand-over
    points -- no steady iteration at all, a cap below / at / just above min_iter, a stop rule that fires at once --
    gives the oracle's iteration count, log-likelihood and rates."""
    from colate_amd import workloads

    grid = ol.age_grid()
    ep, _ = ol.epochs_from_bins(bins)
    csh, cns = workloads.bootstrap_tables(grid, 3, nb=9, scale=1.0, seed=5)
    csh[1, 70:] = 0.0
    cns[1, 70:] = 0.0  # a replicate whose data fit one bin group (its leader keeps the verdict's history itself)
    for max_iter, min_iter in ((1, 1000), (2, 1000), (3, 1), (40, 0), (40, 39), (40, 40), (41, 40), (300, 5), (1200, 1000)):
        kw = dict(max_iter=max_iter, min_iter=min_iter, rel_tol=1e-3 if min_iter < 10 else 1e-7)
        r0, it0, ll0, fl0 = ol.em_batch(grid, csh, cns, ep, **kw)
        r1, it1, ll1, fl1 = ca.em_batch(grid, csh, cns, ep, **kw)
        assert (it0 == it1).all(), (kw, it0, it1)
        assert (fl0 == ca.status_flags(fl1)).all(), (kw, fl0, fl1)
        assert np.allclose(ll1, ll0, rtol=1e-11, atol=0), kw
        mask = ol.stable_mask(grid, csh, cns, ep, r0, **kw)
        # (the checker's stable fraction, measured: 0.971 at 23 epochs, 0.844 .. 0.852 at 122)
        assert mask.mean() > (0.96 if ep.size < 64 else 0.84) and _rel(r1, r0)[mask].max() < RATE_RTOL, (kw, mask.mean(), _rel(r1, r0)[mask].max())


def test_properties_at_baseline_sizes(ca):
    """Size-independent properties at configs[2]/[4] scale (1000 and 2000 replicates in one launch)."""
    from colate_amd import workloads

    grid = ol.age_grid()
    ep, _ = ol.epochs_from_bins("3,7,0.2")
    csh, cns = workloads.bootstrap_tables(grid, 1000)
    r, it, ll, fl = ca.em_batch(grid, csh, cns, ep)
    assert (fl == 0).all() and (it >= 1001).all() and np.isfinite(r).all() and (r >= 0).all()
    # determinism: the same launch twice is bit-identical
    r2, it2, ll2, _ = ca.em_batch(grid, csh, cns, ep)
    assert np.array_equal(r, r2) and np.array_equal(ll, ll2) and (it == it2).all()
    # replicates are independent: any permutation / duplication of rows permutes / duplicates results bit for bit
    perm = np.random.default_rng(0).permutation(1000)
    big_sh, big_ns = np.concatenate([csh[perm], csh]), np.concatenate([cns[perm], cns])
    rp, itp, llp, _ = ca.em_batch(grid, big_sh, big_ns, ep)
    assert np.array_equal(rp[:1000], r[perm]) and np.array_equal(rp[1000:], r) and (itp[:1000] == it[perm]).all()
    # spot-check against the oracle
    idx = [0, 499, 999]
    r0, it0, ll0, _ = ol.em_batch(grid, csh[idx], cns[idx], ep)
    assert (it[idx] == it0).all() and _rel(r[idx], r0).max() < 1e-8
    # scaling every count by 2 is exact in binary: same rates bit for bit, log-likelihood doubled
    rs, its, lls, _ = ca.em_batch(grid, 2 * csh[:8], 2 * cns[:8], ep)
    assert np.array_equal(rs, r[:8]) and np.array_equal(lls, 2 * ll[:8])


def test_device_entry_points_with_torch(ca):
    """colate_em_batch_device / colate_em_estep_device on tensors resident in HBM, on a side stream,
    with per-replicate epochs and starting rates (batched all-pairs layout)."""
    import torch
    from colate_amd import workloads

    grid = ol.age_grid()
    ep_a, _ = ol.epochs_from_bins("3,7,0.2")
    ep_b = ep_a.copy()
    ep_b[2:-1] *= 1.1
    csh, cns = workloads.bootstrap_tables(grid, 4, nb=9, scale=1.0)
    dev = torch.device("cuda")
    f64 = dict(dtype=torch.float64, device=dev)
    eps = torch.tensor(np.stack([ep_a, ep_b, ep_a, ep_b]), **f64)
    init = torch.full((4, ep_a.size), 1.0 / 20000.0, **f64)
    out = torch.empty((4, ep_a.size), **f64)
    it = torch.empty(4, dtype=torch.int32, device=dev)
    ll = torch.empty(4, **f64)
    fl = torch.empty(4, dtype=torch.int32, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ca.em_batch_device(torch.tensor(grid, **f64), torch.tensor(csh, **f64), torch.tensor(cns, **f64), eps, init,
                           out, it, ll, fl, stream=s)
    s.synchronize()
    for b, e in enumerate((ep_a, ep_b, ep_a, ep_b)):
        r0, it0, _, _ = ol.em_batch(grid, csh[b:b + 1], cns[b:b + 1], e)
        assert int(it[b]) == it0[0] and _rel(out[b].cpu().numpy(), r0[0]).max() < 1e-8


def test_sharded_entry_point_matches_single_launch(ca):
    """colate_em_batch_sharded (one process, several shards/streams; here all on GPU 0) returns the
    replicates in order, bit-identical to one launch."""
    from colate_amd import workloads

    grid = ol.age_grid()
    ep, _ = ol.epochs_from_bins("3,7,0.2")
    csh, cns = workloads.bootstrap_tables(grid, 11, nb=9, scale=1.0)
    r1, it1, ll1, fl1 = ca.em_batch(grid, csh, cns, ep)
    for devs in ([0], [0, 0, 0], [0] * 16):
        r, it, ll, fl = ca.em_batch_sharded(devs, grid, csh, cns, ep)
        assert np.array_equal(r, r1) and np.array_equal(ll, ll1) and (it == it1).all() and (fl == fl1).all()
    with pytest.raises(ca.ColateError):
        ca.em_batch_sharded([99], grid, csh, cns, ep)


def test_throughput_variant_is_bit_identical(ca):
    """The kernel has a latency variant (a wave per role and bin group; B up to twice the CU count; built twice:
    max-ilp scheduling -- the one the library picks -- and default scheduling, kept for A/B runs) and a
    throughput variant (two waves per replicate walking through the bin groups; larger B, or more than 128 epochs).  Same phases, same
    arithmetic: rates, log-likelihoods, iteration counts and flags agree bit for bit, so results do not depend
    on the batch size a replicate happens to be run in."""
    from colate_amd import workloads

    grid = ol.age_grid()
    cases = [(ol.epochs_from_bins(bins)[0], nrep) for bins, nrep in (("3,7,0.2", 12), ("2,7.95,0.05", 3), ("3,7,0.3", 3))]
    # both sides of the limits of the 1-, 2- and 4-row instantiations (16 / 32 epochs)
    cases += [(np.concatenate([[0.0], np.geomspace(30, 3e5, n - 2), [4e6]]), 3) for n in (16, 31, 32, 33)]
    for ep, nrep in cases:
        csh, cns = workloads.bootstrap_tables(grid, nrep, nb=9, scale=1.0)
        csh[0, :] = 0.0          # a replicate without shared counts
        cns[1, 60:] = 0.0        # one whose data stop early (one bin group)
        out = {}
        for variant in ("latency-ilp", "latency", "throughput"):  # (the two latency builds differ in scheduling only)
            ca.em_force_variant(variant)
            try:
                assert ca.em_kernel_variant(nrep, ep.size) == variant
                out[variant] = ca.em_batch(grid, csh, cns, ep)
            finally:
                ca.em_force_variant(None)
        for other in ("latency", "throughput"):
            for a, b in zip(out["latency-ilp"], out[other]):
                assert np.array_equal(a, b)
    # the library picks the throughput variant by itself for a batch beyond 2 x #CUs: spot-check against the oracle
    ep, _ = ol.epochs_from_bins("3,7,0.2")
    csh, cns = workloads.bootstrap_tables(grid, 600, nb=115, scale=11.0, seed=3)
    r, it, ll, fl = ca.em_batch(grid, csh, cns, ep)
    idx = [0, 299, 599]
    r0, it0, ll0, _ = ol.em_batch(grid, csh[idx], cns[idx], ep)
    assert (fl == 0).all() and (it[idx] == it0).all() and _rel(r[idx], r0).max() < 1e-8


def test_rows_sharded_entry_point_matches_rows_launch(ca):
    """colate_em_batch_rows_sharded: per-row epochs (batched pairs) sharded over devices/streams; rows come
    back in order, bit-identical to colate_em_batch_rows."""
    from colate_amd import workloads

    grid = ol.age_grid()
    ep_a, _ = ol.epochs_from_bins("3,7,0.2")
    ep_b = ep_a.copy()
    ep_b[2:-1] *= 1.07
    csh, cns = workloads.bootstrap_tables(grid, 7, nb=9, scale=1.0)
    eps = np.stack([ep_a, ep_b, ep_b, ep_a, ep_a, ep_b, ep_a])
    init = np.full(eps.shape, 1.0 / 20000.0)
    init[3] *= 2.0
    r1, it1, ll1, fl1 = ca.em_batch_rows(grid, csh, cns, eps, init)
    for devs in ([0], [0, 0], [0] * 9):
        r, it, ll, fl = ca.em_batch_rows_sharded(devs, grid, csh, cns, eps, init)
        assert np.array_equal(r, r1) and np.array_equal(ll, ll1) and (it == it1).all() and (fl == fl1).all()
    bad = eps.copy()
    bad[5, 4] = bad[5, 3] - 1.0  # a decreasing epoch grid in one row is rejected for the whole call
    with pytest.raises(ca.ColateError):
        ca.em_batch_rows_sharded([0, 0], grid, csh, cns, bad, init)


def test_bootstrap_em_batch_groups_equals_separate_calls(ca):
    """colate_bootstrap_em_batch_groups (batched all-pairs): G pairs with their own block tables,
    numbers of blocks, sample ages and epochs in ONE bootstrap launch + ONE EM launch -- every row bit-identical to the
    per-pair call colate_bootstrap_em_batch, and the count tables bit-identical to the host twin and to the oracle's
    restatement of coal.cpp:3358-3451."""
    rng = np.random.default_rng(17)
    grid = ol.age_grid()
    A, B = grid.size, 5
    groups = []
    for g, (nb, age_years) in enumerate([(9, 0.0), (23, 7000.0), (1, 0.0), (115, 1200.0), (16, 0.0)]):
        prof = (1 - np.exp(-grid / 9000.0))
        sh = rng.uniform(0, 3, (nb, A)) * (rng.uniform(size=(nb, A)) < 0.6) * prof
        ns = rng.uniform(0, 9, (nb, A)) * (rng.uniform(size=(nb, A)) < 0.6)
        she = rng.uniform(0, 1, (nb, A)) * (rng.uniform(size=(nb, A)) < 0.2)
        nse = rng.uniform(0, 1, (nb, A)) * (rng.uniform(size=(nb, A)) < 0.2)
        age = age_years / 28.0
        ep, _ = ol.epochs_from_bins("3,7,0.2", age, 28.0)
        w = ca.bootstrap_weights(ca.Rng(100 + g), B, nb)
        groups.append(dict(age=age, tabs=(sh, ns, she, nse), ep=ep, w=w))
    E = {len(g["ep"]) for g in groups}
    assert len(E) == 1  # (--bins 3,7,0.2: an ancient sample replaces epochs, the count stays)
    out = ca.bootstrap_em_batch_groups(grid, [g["age"] for g in groups], [g["w"] for g in groups], [g["tabs"] for g in groups],
                                       np.stack([g["ep"] for g in groups]), want_counts=True, max_iter=1200)
    rates, iters, ll, flags, csh, cns = out
    for k, g in enumerate(groups):
        r1, it1, ll1, fl1, csh1, cns1 = ca.bootstrap_em_batch(grid, g["age"], g["w"], *g["tabs"], g["ep"], want_counts=True, max_iter=1200)
        rows = slice(k * B, (k + 1) * B)
        assert np.array_equal(csh[rows], csh1) and np.array_equal(cns[rows], cns1)
        assert np.array_equal(rates[rows], r1) and np.array_equal(ll[rows], ll1)
        assert (iters[rows] == it1).all() and (flags[rows] == fl1).all()
        h_sh, h_ns = ca.bootstrap_counts_from_weights(grid, g["age"], g["w"], *g["tabs"])
        assert np.array_equal(csh1, h_sh) and np.array_equal(cns1, h_ns)
        r0, it0, _, fl0 = ol.em_batch(grid, csh1, cns1, g["ep"], max_iter=1200)
        assert (it0 == it1).all() and _rel(r1, r0).max() < 1e-6
    with pytest.raises(ca.ColateError):  # a decreasing epoch grid in one group refuses the whole call
        bad = np.stack([g["ep"] for g in groups])
        bad[3, 5] = bad[3, 4] - 1.0
        ca.bootstrap_em_batch_groups(grid, [g["age"] for g in groups], [g["w"] for g in groups], [g["tabs"] for g in groups], bad)


def test_pairs_mode_one_launch_equals_separate_runs(ca, tmp_path):
    """Batched all-pairs front end: one process, one launch per distinct epoch count; every pair's
    .coal is byte-identical to the .coal of that pair run on its own."""
    gl.l3_stage("l3_ancient", str(tmp_path))
    common = ["--mode", "mut", "--mut", "P", "--chr", "chr.txt", "--bins", "3,7,0.2", "--seed", "5", "--num_bootstraps", "2"]
    specs = [("T.colate.in", "R.colate.in", "ab", "7000", "0"), ("R.colate.in", "T.colate.in", "ba", "0", "0"),
             ("T.colate.in", "T.colate.in", "aa", "0", "0")]
    (tmp_path / "pairs.txt").write_text("".join(" ".join(sp) + "\n" for sp in specs))
    r = subprocess.run([CLI] + common + ["--pairs", "pairs.txt"], cwd=str(tmp_path), capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-800:]
    for tgt, ref, out, ta, ra in specs:
        r = subprocess.run([CLI] + common + ["--target_tmp", tgt, "--reference_tmp", ref, "--target_age", ta,
                                             "--reference_age", ra, "-o", out + "_single"], cwd=str(tmp_path), capture_output=True)
        assert r.returncode == 0, r.stderr.decode()[-800:]
        assert (tmp_path / (out + ".coal")).read_text() == (tmp_path / (out + "_single.coal")).read_text()
    # --devices N shards the (pair, replicate) rows over GPUs 0..N-1: same files
    for out in ("ab", "ba", "aa"):
        (tmp_path / (out + ".coal")).rename(tmp_path / (out + "_one.coal"))
    r = subprocess.run([CLI] + common + ["--pairs", "pairs.txt", "--devices", "1"], cwd=str(tmp_path), capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-800:]
    for out in ("ab", "ba", "aa"):
        assert (tmp_path / (out + ".coal")).read_text() == (tmp_path / (out + "_one.coal")).read_text()


@pytest.mark.parametrize("A", [1, 2, 64, 65, 130, 256])
def test_custom_age_grids(ca, A):
    """The ABI takes any non-decreasing age grid up to 256 bins (the reference reads its grid from
    .colate_mat, coal.cpp:3481-3483): 1..4 bin groups per role, padding lanes, ties on epoch starts."""
    rng = np.random.default_rng(A)
    ep, _ = ol.epochs_from_bins("3,6.5,0.5")
    grid = np.sort(np.exp(rng.uniform(np.log(2.0), np.log(4e5), A)))
    if A >= 64:
        grid[5] = ep[3]   # an age exactly on an epoch start belongs to the epoch that starts there
        grid[6] = ep[3]
        grid = np.sort(grid)
    csh = rng.uniform(0, 30, (3, A)) * (rng.uniform(size=(3, A)) < 0.7) * (1 - np.exp(-grid / 9000.0))
    cns = rng.uniform(0, 60, (3, A)) * (rng.uniform(size=(3, A)) < 0.7)
    r, it, ll, fl = ca.em_batch(grid, csh, cns, ep, max_iter=1300)
    r0, it0, ll0, fl0 = ol.em_batch(grid, csh, cns, ep, max_iter=1300)
    ok = (fl0 & 3) == 0  # (the cap of 1300 iterations may be hit: then both stop there, flagged alike)
    assert ok.any() and (ca.status_flags(fl)[ok] == fl0[ok]).all()
    assert (it[ok] == it0[ok]).all() and np.allclose(ll[ok], ll0[ok], rtol=1e-11, atol=1e-13)
    m = ol.stable_mask(grid, csh, cns, ep, r0, max_iter=1300)
    assert _rel(r, r0)[m & ok[:, None]].max() < RATE_RTOL


def test_flags(ca):
    grid = ol.age_grid()
    ep, _ = ol.epochs_from_bins("3,7,0.2")
    from colate_amd import workloads

    csh, cns = workloads.bootstrap_tables(grid, 2, nb=9, scale=1.0)
    # iteration cap reached before the stop rule can fire
    r, it, ll, fl = ca.em_batch(grid, csh, cns, ep, max_iter=50)
    assert (it == 50).all() and (ca.status_flags(fl) == ca.FLAG_MAXITER).all()
    r0, it0, ll0, fl0 = ol.em_batch(grid, csh, cns, ep, max_iter=50)
    assert _rel(r, r0).max() < 1e-9 and np.allclose(ll, ll0, rtol=1e-12)
    # a not-shared count inside the last epoch with a zero last rate: the reference asserts (coal_EM.cpp:351)
    cns2 = cns.copy()
    cns2[:, 182] = 5.0
    init = np.full(ep.size, 1.0 / 20000.0)
    init[-1] = 0.0
    num, den, ll, fl = ca.em_estep(grid, csh, cns2, ep, np.tile(init, (2, 1)))
    assert (fl & ca.FLAG_NAN).all()


@pytest.mark.parametrize("bins", ["3,7,0.2", "2,7.95,0.05"])
def test_zero_starting_rates_and_failing_normalisers(ca, bins):
    """Starting rates of 0 (a --coal file may hold them): a zero rate is a fixed point of the M-step where nothing older
    coalesces into the epoch, the last epoch stops absorbing (the kernel's cold path), and shared bins inside a run of
    leading zero-rate epochs have a normaliser of 0 -- the reference drops them (coal_EM.cpp:288-292), the kernel
    publishes them to its role leader every iteration.  Same iterations, flags, log-likelihood and rates as the oracle."""
    from colate_amd import workloads

    grid = ol.age_grid()
    ep, _ = ol.epochs_from_bins(bins)
    E = ep.size
    csh, cns = workloads.bootstrap_tables(grid, 2, nb=9, scale=1.0, seed=5)
    third = E // 3
    for zero in ([0], [0, 1, 2], list(range(third)), [third], list(range(third, third + 4)), [E - 1], [0, E - 1], [E - 2, E - 1]):
        init = np.full(E, 1.0 / 20000.0)
        init[zero] = 0.0
        kw = dict(init=init, max_iter=60, min_iter=10)
        r0, it0, ll0, fl0 = ol.em_batch(grid, csh, cns, ep, **kw)
        if (fl0 & 3).any():  # (the reference itself aborts on this start)
            continue
        r1, it1, ll1, fl1 = ca.em_batch(grid, csh, cns, ep, init_rates=init, max_iter=60, min_iter=10)
        assert (it0 == it1).all() and (fl0 == ca.status_flags(fl1)).all(), (zero, it0, it1, fl0, fl1)
        assert np.allclose(ll1, ll0, rtol=1e-11, atol=0), (zero, ll0, ll1)
        assert ((r0 == 0) == (r1 == 0)).all(), zero
        mask = ol.stable_mask(grid, csh, cns, ep, r0, **kw)
        # (measured: 1.0 at 23 epochs, 0.861 at 122)
        assert mask.mean() > (0.99 if ep.size < 64 else 0.85) and _rel(r1, r0)[mask].max() < RATE_RTOL, (zero, mask.mean(), _rel(r1, r0)[mask].max())


def test_device_bootstrap_bit_identical_to_host(ca):
    """Block bootstrap on the GPU (weighted block sums + F redistribution, coal.cpp:3358-3441) against
    the host path and against the oracle, same std::mt19937 weights: bit-identical tables, then EM straight from HBM."""
    import torch

    rng = np.random.default_rng(5)
    grid = ol.age_grid()
    A, nb, B = grid.size, 115, 64
    sh = rng.uniform(0, 3, (nb, A)) * (rng.uniform(size=(nb, A)) < 0.6)
    ns = rng.uniform(0, 9, (nb, A)) * (rng.uniform(size=(nb, A)) < 0.6)
    she = rng.uniform(0, 1, (nb, A)) * (rng.uniform(size=(nb, A)) < 0.2)
    nse = rng.uniform(0, 1, (nb, A)) * (rng.uniform(size=(nb, A)) < 0.2)
    sh[:, :30] = ns[:, :30] = she[:, :30] = nse[:, :30] = 0
    dev = torch.device("cuda")
    f64 = dict(dtype=torch.float64, device=dev)
    for age in (0.0, 250.0):
        for nboot in (1, B):
            h_sh, h_ns = ca.bootstrap_counts(ca.Rng(777), nboot, grid, age, sh, ns, she, nse)
            w = ca.bootstrap_weights(ca.Rng(777), nboot, nb)
            d_sh = torch.empty((nboot, A), **f64)
            d_ns = torch.empty((nboot, A), **f64)
            status = ca.bootstrap_counts_device(torch.tensor(grid, **f64), age, torch.tensor(w, **f64), torch.tensor(sh, **f64),
                                                torch.tensor(ns, **f64), torch.tensor(she, **f64), torch.tensor(nse, **f64),
                                                d_sh, d_ns)
            torch.cuda.synchronize()
            assert int(status.item()) == 0
            assert np.array_equal(d_sh.cpu().numpy(), h_sh) and np.array_equal(d_ns.cpu().numpy(), h_ns)
            # ... and against the ORACLE directly (its own mt19937 + block weights + weighted sums + F redistribution)
            import ctypes

            g = (ctypes.c_uint * 625)()
            ol.O.oracle_mt_seed(g, 777)
            for i in range(nboot):
                wo = np.zeros(nb)
                ol.O.oracle_block_weights(g, nb, nboot, ol.P(wo))
                o_sh, o_ns = np.zeros(A), np.zeros(A)
                ol.O.oracle_bootstrap_counts(nb, A, ol.P(grid), age, ol.P(wo), ol.P(sh), ol.P(ns), ol.P(she), ol.P(nse), ol.P(o_sh), ol.P(o_ns))
                assert np.array_equal(d_sh[i].cpu().numpy(), o_sh) and np.array_equal(d_ns[i].cpu().numpy(), o_ns)
    # and on into the EM without leaving HBM
    ep, _ = ol.epochs_from_bins("3,7,0.2")
    E = ep.size
    out = torch.empty((B, E), **f64)
    it = torch.empty(B, dtype=torch.int32, device=dev)
    ll = torch.empty(B, **f64)
    fl = torch.empty(B, dtype=torch.int32, device=dev)
    ca.em_batch_device(torch.tensor(grid, **f64), d_sh, d_ns, torch.tensor(ep, **f64),
                       torch.full((E,), ca.DEFAULT_INIT_RATE, **f64), out, it, ll, fl, max_iter=60)
    r0, it0, _, _ = ca.em_batch(grid, h_sh, h_ns, ep, max_iter=60)
    assert np.array_equal(out.cpu().numpy(), r0)


def test_loglik_trace_and_profiler_ranges(ca, tmp_path, monkeypatch):
    """COLATE_LL_TRACE=<file>: the log-likelihood of every iteration (the reference's commented-out trace, coal.cpp:3659,
    3817): same rates, iterations and final log-likelihood as the untraced run, one line per executed E-step, and the
    sequence increases like an EM's.  COLATE_ROCTX=1 (roctx ranges around the host-pointer calls) must not change anything."""
    from colate_amd import workloads

    grid = ol.age_grid()
    ep, _ = ol.epochs_from_bins("3,7,0.2")
    csh, cns = workloads.bootstrap_tables(grid, 3, nb=9, scale=1.0)
    kw = dict(max_iter=300, min_iter=100)
    r0, it0, ll0, fl0 = ca.em_batch(grid, csh, cns, ep, **kw)
    path = tmp_path / "ll.txt"
    monkeypatch.setenv("COLATE_LL_TRACE", str(path))
    monkeypatch.setenv("COLATE_ROCTX", "1")
    r1, it1, ll1, fl1 = ca.em_batch(grid, csh, cns, ep, **kw)
    monkeypatch.delenv("COLATE_LL_TRACE")
    assert np.array_equal(r0, r1) and np.array_equal(it0, it1) and np.array_equal(ll0, ll1) and np.array_equal(fl0, fl1)
    rows = np.loadtxt(path)
    for b in range(3):
        tr = rows[rows[:, 0] == b]
        n_esteps = min(int(it0[b]) + 1, 300)
        assert len(tr) == n_esteps and (tr[:, 1] == np.arange(n_esteps)).all()
        assert tr[-1, 2] == ll0[b]
        assert (np.diff(tr[:, 2]) >= -1e-9 * np.abs(tr[1:, 2])).all()  # the EM's log-likelihood does not decrease
