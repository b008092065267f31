"""Loaders for the committed golden fixtures (tests/golden/, generated from the reference by make_golden.py)."""
import gzip
import json
import os
import shutil

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def unhex(xs):
    return np.array([float.fromhex(x) for x in xs])


def l1_cases():
    import oracle_lib as ol

    data = json.load(open(os.path.join(HERE, "l1_estep.json")))
    cache = {}
    for c in data["cases"]:
        if "bins" in c:
            if c["bins"] not in cache:
                cache[c["bins"]] = ol.epochs_from_bins(c["bins"])[0]
            ep = cache[c["bins"]]
        else:
            ep = unhex(c["epochs"])
        yield dict(epochs=ep, kind=c["kind"], age=float.fromhex(c["age"]), rates=unhex(c["rates"]),
                   logl=float.fromhex(c["logl"]), num=unhex(c["num"]), denom=unhex(c["denom"]))


def l2_names():
    return sorted(f[len("l2_em_"):-5] for f in os.listdir(HERE) if f.startswith("l2_em_"))


def l2_case(name):
    d = json.load(open(os.path.join(HERE, f"l2_em_{name}.json")))
    return dict(bins=d["bins"], csh=np.array([unhex(r) for r in d["cnt_shared"]]),
                cns=np.array([unhex(r) for r in d["cnt_notshared"]]), coal=d["coal"], iterations=d["iterations"])


def l3_names():
    return sorted(f for f in os.listdir(HERE) if f.startswith("l3_") and f != "l3_pairs")


def l3_pairs_stage(dst):
    """The batched all-pairs fixture (l3_pairs: the reference run once per pair): copies the inputs and the expected
    .coal files into `dst`, writes pairs.txt (`target reference output target_age reference_age` per line) and returns
    the case description."""
    src = os.path.join(HERE, "l3_pairs")
    os.makedirs(dst, exist_ok=True)
    for f in os.listdir(src):
        if f.endswith(".colate.in.gz"):
            with gzip.open(os.path.join(src, f), "rb") as g, open(os.path.join(dst, f[:-3]), "wb") as o:
                o.write(g.read())
        else:
            shutil.copy(os.path.join(src, f), os.path.join(dst, f))
    meta = json.load(open(os.path.join(src, "case.json")))
    with open(os.path.join(dst, "pairs.txt"), "w") as f:
        for p in meta["pairs"]:
            f.write(f"{p['target']} {p['reference']} {p['output']} {p['target_age']} {p['reference_age']}\n")
    return meta


def l3_stage(name, dst):
    """Copy an L3 case into `dst`, decompressing the .colate.in files (the reference freads them raw)."""
    src = os.path.join(HERE, name)
    os.makedirs(dst, exist_ok=True)
    meta = json.load(open(os.path.join(src, "case.json")))
    if "inputs_from" in meta:  # this case runs on another case's input files (stored once)
        l3_stage(meta["inputs_from"], dst)
        os.remove(os.path.join(dst, "expected.coal"))
    for f in os.listdir(src):
        if f.endswith(".colate.in.gz"):
            with gzip.open(os.path.join(src, f), "rb") as g, open(os.path.join(dst, f[:-3]), "wb") as o:
                o.write(g.read())
        else:
            shutil.copy(os.path.join(src, f), os.path.join(dst, f))
    return json.load(open(os.path.join(src, "case.json")))


def coal_text(epochs, rates, is_ancient=False, ep_null=0):
    """The reference's .coal format (coal.cpp:3660-3672, 3830-3847); `%g` == default ostream << double."""
    E = len(epochs)
    out = "0\n"
    if is_ancient:
        out += "0 " + "".join("%g " % epochs[e] for e in range(ep_null + 1, E)) + "\n"
    else:
        out += "".join("%g " % x for x in epochs) + "\n"
    for i, r in enumerate(np.atleast_2d(rates)):
        if is_ancient:
            out += "0 %d " % i + "".join("%g " % (0.0 if e <= ep_null else r[e]) for e in range(ep_null, E)) + "\n"
        else:
            out += "0 %d " % i + "".join("%g " % x for x in r) + "\n"
    return out


def read_counts(path, B, A=185):
    """Count tables in the .colate_mat layout (what `Colate --counts_out` writes)."""
    v = np.array(open(path).read().split(), dtype=np.float64)
    grid = v[:A]
    rest = v[A:].reshape(B, 2, A)
    return grid, rest[:, 0, :].copy(), rest[:, 1, :].copy()


def write_colate_mat(path, grid, csh, cns):
    """The reference's .colate_mat layout (coal.cpp:3481-3497), 17 significant digits as make_golden.py wrote it."""
    with open(path, "w") as f:
        f.write(" ".join("%.17g" % x for x in grid) + "\n")
        for b in range(len(csh)):
            f.write(" ".join("%.17g" % x for x in csh[b]) + "\n")
            f.write(" ".join("%.17g" % x for x in cns[b]) + "\n")
