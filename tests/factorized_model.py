"""numpy model of the *factorised* E-step that the HIP kernel implements.

TEST INFRASTRUCTURE.  The reference (coal_EM.cpp:153-468 called per age bin from
coal.cpp:3704-3733) evaluates, for every (age bin, epoch) pair, exp(log-term - Z)
in the log domain: O(A*E) transcendentals per EM iteration.  The kernel computes
the same sufficient statistics N_e = sum_b c_b num_e(b), D_e = sum_b c_b denom_e(b)
and ll = sum_b c_b Z_b in the linear domain, in O(A + E), by factoring every
per-(bin, epoch) term into (per-epoch) x (per-bin) pieces.  DESIGN.md derives it;
this file is the executable form of that derivation, kept line-for-line parallel
to colate_amd/csrc/em_kernels.hip so that the algebra can be checked against
oracle/ on the CPU (tests/test_factorized_model.py) without a GPU.
"""
import numpy as np


# Round 2's mean model of the reference's `integ` rounding residue, per unit count.  The kernel keeps it for the shared kind only;
# for the not-shared kind it now carries the tail model of DESIGN.md section 6 (absorbed log-sum-exp terms, clamp, rounding noise),
# whose executable CPU statement is tools/study/residue_models.cpp (model 5).  On the tables of tests/test_factorized_model.py
# (one E-step at well-conditioned rates) the two are indistinguishable.
INTEG_RESIDUE = 4.0e-17


def epoch_index(age_grid, epochs):
    """k(a) = largest e with epochs[e] <= a  (coal_EM.cpp:66 strict `age < epochs[e]`)."""
    return np.searchsorted(epochs, age_grid, side="right") - 1


def estep(epochs, rates, age_grid, c_sh, c_ns):
    """One factorised E-step.  Returns (N[E], D[E], ll)."""
    t = np.asarray(epochs, dtype=np.float64)
    lam = np.asarray(rates, dtype=np.float64)
    E = t.size
    A = age_grid.size
    kb = epoch_index(age_grid, t)
    with np.errstate(all="ignore"):
        # ---- epoch pass (get_AB, coal_EM.cpp:97-151, in the linear domain) ----
        dt = np.zeros(E)
        dt[:-1] = t[1:] - t[:-1]
        x = lam * dt
        cs = np.zeros(E + 1)
        for e in range(E - 1):
            cs[e + 1] = cs[e] + x[e]  # sequential, as coal_EM.cpp:100-103
        inv = 1.0 / lam
        q = np.exp(-cs[1:] + cs[:-1])  # q_e = exp(-cs_{e+1} + cs_e); q_{E-1} unused
        S = np.exp(-cs[:E])
        valid = np.zeros(E, dtype=bool)
        valid[:-1] = (lam[:-1] > 0) & (t[1:] != 0) & (dt[:-1] > 0)
        valid[-1] = lam[-1] > 0
        p = np.where(valid, 1.0 - q, 0.0)
        beta = np.where(valid, (t + inv) - (np.append(t[1:], 0.0) + inv) * q, 0.0)
        p[-1] = 1.0 if valid[-1] else 0.0
        beta[-1] = (t[-1] + inv[-1]) if valid[-1] else 0.0
        q[-1] = 0.0
        W = S * p          # exp(A_ep)
        V = S * beta       # exp(B_ep)
        VW = V - t * W
        PW = np.zeros(E + 1)
        for e in range(E):
            PW[e + 1] = PW[e] + W[e]
        G = np.zeros(E + 1)  # G_e = sum_{j>=e} W_j / S_e, backward recurrence
        G[E - 1] = p[E - 1]
        for e in range(E - 2, -1, -1):
            G[e] = p[e] + q[e] * G[e + 1]
        Xa = (t + inv) / inv

        # ---- bin pass ----
        g = np.zeros(E); gc = np.zeros(E); gW = np.zeros(E); gV = np.zeros(E)
        h = np.zeros(E); hc = np.zeros(E); hN = np.zeros(E); hD = np.zeros(E)
        ll = 0.0
        for b in range(A):
            k = kb[b]
            a = age_grid[b]
            lk = lam[k]
            ck = cs[k]
            ck1 = ck + lk * (a - t[k])
            if c_sh[b] > 0:
                c = c_sh[b]
                if lk > 0:
                    qd = np.exp(-ck1 + ck)
                    Wp = S[k] * (1.0 - qd)
                    X = Xa[k] - (a + inv[k]) / inv[k] * qd
                    Vp = X * inv[k] * S[k]
                else:
                    Wp = 0.0
                    Vp = 0.0
                Sig = PW[k] + Wp
                if Sig > 0 and np.isfinite(Sig):
                    r = 1.0 / Sig
                    ll += c * np.log(Sig)
                    nk = Wp * r
                    dk = Vp * r - t[k] * nk
                    g[k] += c * r
                    gc[k] += c
                    gW[k] += c * nk
                    gV[k] += c * max(dk, 0.0)
            if c_ns[b] > 0:
                c = c_ns[b]
                ck2 = ck1 + lk * (a - a)
                if k < E - 1:
                    if lk > 0:
                        ck3 = ck2 + lk * (t[k + 1] - a)
                        u = np.exp(-ck3 + ck2)
                        pn = 1.0 - u
                        bn = (a + inv[k]) - (t[k + 1] + inv[k]) * u
                    else:
                        u = 1.0
                        pn = 0.0
                        bn = 0.0
                    Sig = pn + u * G[k + 1]
                    if Sig > 0 and np.isfinite(Sig):
                        rr = 1.0 / Sig
                        ll += c * (-ck2 + np.log(Sig))
                        nk = pn * rr
                        dk = bn * rr - t[k] * nk + dt[k] * (1.0 - nk)
                        h[k] += c * (u * rr)
                        hc[k] += c
                        hN[k] += c * nk
                        hD[k] += c * max(dk, 0.0)
                else:
                    ll += c * (-ck2)
                    hc[k] += c
                    hN[k] += c
                    hD[k] += c * max((a + inv[k]) - t[k], 0.0)

        # ---- epoch accumulation ----
        RS = np.zeros(E + 1); CS = np.zeros(E + 1); CN = np.zeros(E + 1)
        for e in range(E - 1, -1, -1):
            RS[e] = RS[e + 1] + g[e]
            CS[e] = CS[e + 1] + gc[e]
            CN[e] = CN[e + 1] + hc[e]
        T = np.zeros(E + 1)
        for e in range(E - 1):
            T[e + 1] = q[e] * T[e] + h[e]
        N = np.zeros(E); D = np.zeros(E)
        for e in range(E):
            rs, cs_, cn = RS[e + 1], CS[e + 1], CN[e + 1]  # bins in later epochs only
            N[e] = W[e] * rs + gW[e] + p[e] * T[e] + hN[e]
            if e < E - 1:
                D[e] = (VW[e] * rs + dt[e] * (cs_ - PW[e + 1] * rs) + gV[e]
                        + dt[e] * cn + (beta[e] - t[e] * p[e]) * T[e]
                        + dt[e] * G[e + 1] * (q[e] * T[e]) + hD[e])
                # the rounding residue of the reference's `integ` (em_kernels.hip kIntegResidue)
                D[e] += dt[e] * (INTEG_RESIDUE * (c_sh.sum() + c_ns.sum()))
            else:
                D[e] = gV[e] + (beta[e] - t[e] * p[e]) * T[e] + hD[e]
    return N, D, ll


def mstep(N, D, rates, floor=5e-9):
    """coal.cpp:3771-3815 (EM branch)."""
    r = rates.copy()
    for e in range(r.size):
        if N[e] == 0:
            r[e] = r[e - 1] if e > 0 else 0.0
        elif D[e] == 0:
            pass
        else:
            r[e] = max(N[e] / D[e], floor)
    return r
