#!/usr/bin/env python3
"""numpy emulation of binf_root's iteration (csrc/spx_group_common.hpp) on groups of 8, one lane per group, on the bench
distribution (xk ~ N(0,1), sj ~ U(-1/2,1/2), q ~ N(0,1), lambda ~ U(1/2,3/2), sigma = Delta = 1): counts the passes over the
group and the scalar Newton steps of the piece solves, per group and as the maximum over a wavefront of 64 groups (which is
what the wavefront pays), for the piece solvers tried in round 4.  CPU only."""
import sys
import numpy as np

EPS = 2.220446049250313e-16


def ab(S, X, tau, delta):
    z = tau[:, None] * S - X
    act = np.abs(z) > delta
    b = X + np.sign(z) * delta
    sa = np.where(act, 0.0, S * S).sum(1)
    sb = np.where(act, b * b, 0.0).sum(1)
    return sa, sb


def piece_newton_v(sa, sb, sl, u):
    """the kernel's piece solve: Newton in v from u; returns root, steps"""
    v = u.copy()
    steps = np.zeros(len(u), int)
    done = sb == 0.0
    v = np.where(done, np.sqrt(sa) - sl, v)
    for k in range(64):
        live = ~done
        if not live.any():
            break
        t = v / (sl + v)
        ph = np.sqrt(sb + sa * t * t)
        g = v - ph
        gp = 1.0 - np.where(ph > 0, sa * t * (sl / (sl + v) ** 2) / np.where(ph > 0, ph, 1), 0.0)
        vn = v - g / gp
        last = np.abs(vn - v) <= 1e-8 * np.abs(vn)
        steps += live
        v = np.where(live, vn, v)
        done = done | (live & last)
    return v, steps


def poly_coeffs(sa, sb, sl):
    # P(t) = (1 - t)^2 (B + A t^2) - c^2 t^2 = A t^4 - 2A t^3 + (A + B - c^2) t^2 - 2B t + B
    return sa, -2 * sa, sa + sb - sl * sl, -2 * sb, sb


def piece_newton_poly(sa, sb, sl, u, start="cur"):
    """Newton on the quartic in t = v / (c + v); returns root v and steps"""
    c4, c3, c2, c1, c0 = poly_coeffs(sa, sb, sl)
    t = u / (sl + u)
    steps = np.zeros(len(u), int)
    done = sb == 0.0
    for k in range(64):
        live = ~done
        if not live.any():
            break
        P = (((c4 * t + c3) * t + c2) * t + c1) * t + c0
        dP = ((4 * c4 * t + 3 * c3) * t + 2 * c2) * t + c1
        tn = t - P / dP
        last = np.abs(tn - t) <= 1e-8 * np.abs(tn)
        steps += live
        t = np.where(live, tn, t)
        done = done | (live & last)
    v = np.where(sb == 0.0, np.sqrt(sa) - sl, sl * t / (1 - t))
    return v, steps


def run(ngroups, gs, solver, seed=1, start_ns=False, confirm_from=1):
    rng = np.random.default_rng(seed)
    xk = rng.standard_normal((ngroups, gs))
    sj = rng.random((ngroups, gs)) - 0.5
    q = rng.standard_normal((ngroups, gs))
    lam = rng.random(ngroups) + 0.5
    sigma = delta = 1.0
    S = (q + xk) + sj
    X = xk
    sl = lam * sigma
    sS = (S * S).sum(1)
    sX = (X * X).sum(1)
    ub = np.sqrt(sS + sX) * (1 + 8 * EPS)
    ul = sl * EPS
    ulo = np.full(ngroups, 0.0) + ul
    uhi = ub.copy()
    u = ub.copy()
    if start_ns:
        u = np.maximum(np.sqrt(sS) - sl, 1e-3 * ub)
    tau = u / (sl + u)
    sa, sb = ab(S, X, tau, delta)
    psi = u - np.sqrt(sb + tau * tau * sa)
    passes = np.ones(ngroups, int)
    nsteps = np.zeros(ngroups, int)
    wave_steps = np.zeros(ngroups // 64, int)   # sum over iterations of the per-wavefront maximum of the steps
    wave_pieces = np.zeros(ngroups // 64, int)  # iterations in which some lane of the wavefront solved a piece
    pa = np.full(ngroups, -1.0)
    pb = np.full(ngroups, -1.0)
    done = np.zeros(ngroups, bool)
    nS = np.sqrt(sS)
    tau_full = tau.copy()
    wave_conf = np.zeros(ngroups // 64, int)   # iterations in which some lane of the wavefront ran the cheap confirmation
    wave_full = np.ones(ngroups // 64, int)    # ... a full pass (the first one included)
    zero = psi < 0  # froot(lmax) < 0 with fl < 0 -> zeros (only when starting from the bound)
    if not start_ns:
        done |= zero
    for it in range(60):
        conv = (np.abs(psi) <= 4 * EPS * u) | ((sa == pa) & (sb == pb))
        done |= conv
        live = ~done
        if not live.any():
            break
        ulo = np.where(live & (psi < 0), u, ulo)
        uhi = np.where(live & ~(psi < 0), u, uhi)
        v, st = solver(sa, sb, sl, u)
        st = np.where(live, st, 0)
        nsteps += st
        wave_steps += st[: (ngroups // 64) * 64].reshape(-1, 64).max(1)
        wave_pieces += live[: (ngroups // 64) * 64].reshape(-1, 64).any(1)
        same = np.abs(v - u) <= 4 * EPS * np.abs(u)
        done |= live & same
        live &= ~same
        exact = (v > ulo) & (v < uhi)
        v = np.where(exact, v, np.sqrt(ulo) * np.sqrt(uhi))
        pa = np.where(live, np.where(exact, sa, -1.0), pa)
        pb = np.where(live, np.where(exact, sb, -1.0), pb)
        u = np.where(live, v, u)
        tau = u / (sl + u)
        # the cheap confirmation (binf_same_active_set) in place of a pass whose sums would come back identical
        W_ = (ngroups // 64) * 64
        if it >= confirm_from:
            tryc = live & exact & (np.abs(tau - tau_full) * nS <= 2.0 * delta)
            za = tau[:, None] * S - X
            zb = tau_full[:, None] * S - X
            same_set = ((np.abs(za) > delta) == (np.abs(zb) > delta)).all(1)
            wave_conf += tryc[:W_].reshape(-1, 64).any(1)
            done |= tryc & same_set
            live &= ~(tryc & same_set)
        wave_full += live[:W_].reshape(-1, 64).any(1)
        tau_full = np.where(live, tau, tau_full)
        sa2, sb2 = ab(S, X, tau, delta)
        sa = np.where(live, sa2, sa)
        sb = np.where(live, sb2, sb)
        psi = np.where(live, u - np.sqrt(sb + tau * tau * sa), psi)
        passes += live
    W = (ngroups // 64) * 64
    pw = passes[:W].reshape(-1, 64).max(1)
    return dict(passes_mean=passes.mean(), passes_wave=pw.mean(), steps_mean=nsteps.mean(), steps_wave=wave_steps.mean(),
                pieces_wave=wave_pieces.mean(), conf_wave=wave_conf.mean(), full_wave=wave_full.mean(), zero_frac=zero.mean(), u=u)


if __name__ == "__main__":
    ng = int(sys.argv[1]) if len(sys.argv) > 1 else 64 * 2000
    for name, solver in (("newton_v", piece_newton_v), ("newton_poly", piece_newton_poly)):
        for sn in (False, True):
            for cf in (1, 0):
                r = run(ng, 8, solver, start_ns=sn, confirm_from=cf)
                u = r.pop("u")
                print(name, "start_ns" if sn else "start_ub", "confirm_from", cf, {k: round(float(v), 3) for k, v in r.items()})
