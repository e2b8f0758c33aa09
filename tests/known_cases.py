"""DFMC on the known entries only (skf_relation_desc.known_bound, csrc/skf_known.h) against the dense path that keeps the
completed relation -- the SAME cases on the host emulator (small) and on the GPU (large ranks, all list-pass kernels).
The dense path itself is pinned to the reference's goldens elsewhere; the two formulations differ by associativity only
(reference _dfmc.py:287-292, 311-325, 341-352), so in f64 they must agree to rounding."""
import numpy as np

import skfusion_amd._native as nat
from skfusion_amd._engine import DevicePlan
from helpers import relerr, within


def masked_graph(n, ranks, known_share, seed=0):
    """a x b ratings-like relation with `known_share` of its entries known, an unmasked b x c relation, a second masked
    a x c relation (denser), and a constraint on b."""
    rs = np.random.RandomState(seed)
    types = ['a', 'b', 'c']
    Ga = rs.rand(n['a'], 4)
    Gb = rs.rand(n['b'], 4)
    R_ab = (Ga @ rs.rand(4, 4) @ Gb.T) / 4.0 + 0.05 * rs.rand(n['a'], n['b'])
    M_ab = rs.rand(n['a'], n['b']) >= known_share                 # True = unknown
    R_bc = (rs.rand(n['b'], n['c']) < 0.2).astype(np.float64)
    R_ac = rs.rand(n['a'], n['c'])
    M_ac = rs.rand(n['a'], n['c']) >= min(4 * known_share, 0.2)
    theta = -0.01 * (rs.rand(n['b'], n['b']) < 2.0 / n['b'])
    theta = theta + theta.T
    np.fill_diagonal(theta, 0.02)
    rels = [('a', 'b', R_ab, M_ab), ('b', 'c', R_bc, None), ('a', 'c', R_ac, M_ac)]
    thetas = [('b', theta)]
    G0 = {t: rs.rand(n[t], ranks[t]) + 0.1 for t in types}
    return types, rels, thetas, G0


def run(types, n, ranks, rels, thetas, G0, dtype, iters, sparse, with_errors=False):
    plan = DevicePlan(types, n, ranks, rels, thetas, nat.SKF_DFMC, dtype=dtype, sparse_known=None if sparse else False)
    try:
        for t in types:
            plan.set_factor(t, G0[t])
        errs = []
        if with_errors:
            for _ in range(iters):
                plan.iterate(1)
                errs.append([plan.relation_sqerr(k) for k in range(len(rels))])
        else:
            plan.iterate(iters)
        G = {t: plan.get_factor(t) for t in types}
        S = [plan.get_backbone(k) for k in range(len(rels))]
        extra = {}
        if sparse:
            extra['A'] = plan.get_contraction(0, 2)
            extra['Q'] = plan.get_contraction(0, 1)
        else:
            extra['P'] = plan.get_contraction(0, 0)
            extra['Q'] = plan.get_contraction(0, 1)
            extra['S'] = S[0]
        return G, S, np.array(errs), extra
    finally:
        plan.close()


def sparse_against_dense(n, ranks, known_share, iters, dtype, tol, what, monkeypatch, parts=1, seed=0):
    """Same graph, same start: lists of known entries vs completed dense copy.  tol = (G, S, squared errors, P S^T, Q)."""
    tol_g, tol_s, tol_e, tol_a, tol_q = tol
    monkeypatch.setenv('SKF_DFMC_SPARSE', '1')              # whenever a bound is given (up to a quarter known)
    monkeypatch.setenv('SKF_KNOWN_PARTS', str(parts))
    types, rels, thetas, G0 = masked_graph(n, ranks, known_share, seed)
    Gs, Ss, Es, xs = run(types, n, ranks, rels, thetas, G0, dtype, iters, True, with_errors=True)
    Gd, Sd, Ed, xd = run(types, n, ranks, rels, thetas, G0, dtype, iters, False, with_errors=True)
    for t in types:
        within(relerr(Gs[t], Gd[t]), tol_g, '%s: known-entries DFMC vs dense, G_%s after %d iterations' % (what, t, iters))
    for k in range(len(rels)):
        within(relerr(Ss[k], Sd[k]), tol_s, '%s: known-entries DFMC vs dense, S_%d' % (what, k))
    within(np.max(np.abs(Es - Ed) / Ed), tol_e, '%s: known-entries DFMC vs dense, squared errors of every iteration' % what)
    # the row-side product the lists form instead of P:  A = P S^T  (dense path: P and S of the same iteration)
    within(relerr(xs['A'], xd['P'].astype(np.float64) @ xd['S'].T), tol_a, '%s: row-side product P S^T' % what)
    within(relerr(xs['Q'], xd['Q']), tol_q, '%s: column contraction Q' % what)
    return Gs, Ss
