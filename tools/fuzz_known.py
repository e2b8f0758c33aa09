"""Randomised DFMC graphs with masked relations through the device engine -- lists of the known entries (SKF_DFMC_SPARSE=1) and
the completed dense copy (=0) -- against the NumPy oracle:   python tools/fuzz_known.py [n_graphs] [seed]
Sizes 40..500 objects, ranks 2..96 below 3/4 of the objects (row type of the masked relations also 128 / 256), lists in 1 / 2 / 4 / 8 parts, 1-25 % of a masked relation known, rows / columns without a known entry, unmasked
relations beside the masked ones, sparse constraints.  f64: 1e-8 vs the oracle; bf16 / f32: the two device forms against
each other on the reconstruction error.  Exits non-zero on a violation."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import skfusion_amd._native as nat                                    # noqa: E402
from skfusion_amd._engine import DevicePlan                          # noqa: E402
from oracle import dfmf_oracle as orc                                 # noqa: E402
from helpers import relerr                                            # noqa: E402


def random_graph(rs):
    types = ['a', 'b', 'c']
    n = {t: int(rs.choice([40, 64, 65, 130, 257, 400, 500])) for t in types}
    rank = {t: int(rs.choice([2, 5, 16, 31, 32, 48, 64, 96])) for t in types}
    # (rank >= objects makes G^T G singular: the pseudo-inverse's cut-off then decides on rounding noise and ANY two arithmetic
    #  variants drift apart -- seeds with such types failed between the f32 forms as well; the well-posed case is fuzzed)
    rank = {t: min(rank[t], max(2, (3 * n[t]) // 4)) for t in types}
    wide = [r for r in (128, 256) if 2 * r <= n['a']]            # the row type of the masked relations at the widths of
    if wide and rs.rand() < 0.6:                                 # srp_bf16_v6_kernel (one / two 16-byte chunks per lane);
        rank['a'] = int(rs.choice(wide))                         # (rank ~ objects: bf16 noise of EITHER form dominates)
    share = float(rs.choice([0.01, 0.03, 0.1, 0.25]))
    Ga, Gb = rs.rand(n['a'], 4), rs.rand(n['b'], 4)
    R_ab = (Ga @ rs.rand(4, 4) @ Gb.T) / 4.0 + 0.05 * rs.rand(n['a'], n['b'])
    M_ab = rs.rand(n['a'], n['b']) >= share                      # True = unknown
    M_ab[rs.randint(n['a'])] = True                              # a row and a column without a known entry
    M_ab[:, rs.randint(n['b'])] = True
    R = {('a', 'b'): [R_ab], ('b', 'c'): [(rs.rand(n['b'], n['c']) < 0.2).astype(np.float64)]}
    M = {('a', 'b'): [M_ab], ('b', 'c'): [None]}
    if rs.rand() < 0.6:
        R['a', 'c'] = [rs.rand(n['a'], n['c'])]
        M['a', 'c'] = [rs.rand(n['a'], n['c']) >= min(2 * share, 0.2)]
    Theta = {}
    if rs.rand() < 0.6:
        T = -0.01 * (rs.rand(n['b'], n['b']) < 2.0 / n['b'])
        T = T + T.T
        np.fill_diagonal(T, 0.02)
        Theta['b', 'b'] = [T]
    G0 = {(t, t): rs.rand(n[t], rank[t]) + 0.1 for t in types}
    parts = int(rs.choice([0, 0, 2, 4, 8]))                      # 0: the engine's own choice
    return types, n, rank, share, R, M, Theta, G0, parts


def run(types, n, rank, R, M, Theta, G0, dtype, iters, sparse):
    os.environ['SKF_DFMC_SPARSE'] = '1' if sparse else '0'
    rels = [(i, j, m, M[i, j][k]) for (i, j), ms in R.items() for k, m in enumerate(ms)]
    thetas = [(t, T) for (t, _), ms in Theta.items() for T in ms]
    plan = DevicePlan(types, n, rank, rels, thetas, nat.SKF_DFMC, dtype=dtype, sparse_known=None if sparse else False)
    try:
        for t in types:
            plan.set_factor(t, G0[t, t])
        plan.iterate(iters)
        G = {t: plan.get_factor(t) for t in types}
        S = [plan.get_backbone(k) for k in range(len(rels))]
        err = [float(np.sqrt(plan.relation_sqerr(k))) for k in range(len(rels))]
        return G, S, err
    finally:
        plan.close()


def main(n_graphs, seed, iters=4):
    rs = np.random.RandomState(seed)
    bad = 0
    for g in range(n_graphs):
        types, n, rank, share, R, M, Theta, G0, parts = random_graph(rs)
        if parts:
            os.environ['SKF_KNOWN_PARTS'] = str(parts)
        else:
            os.environ.pop('SKF_KNOWN_PARTS', None)
        Go, So = orc.dfmc(R, M, Theta, types, rank, max_iter=iters, G0=G0)
        so = [m for key in R for m in So[key]]
        out = []
        for sparse in (True, False):
            G, S, _ = run(types, n, rank, R, M, Theta, G0, 'f64', iters, sparse)
            w = max([relerr(G[t], Go[t, t]) for t in types] + [relerr(a, b) for a, b in zip(S, so)])
            out.append('f64/%s %.1e' % ('lists' if sparse else 'dense', w))
            bad += not (w < 1e-8)
        over = any(rank[t] > n[t] for t in types)               # rank > objects: the rounding noise of either form, amplified
        for dtype, tol in (('f32', 2e-4), ('bf16', 1.5e-1 if over else 5e-2)):
            _, _, e1 = run(types, n, rank, R, M, Theta, G0, dtype, iters, True)
            _, _, e0 = run(types, n, rank, R, M, Theta, G0, dtype, iters, False)
            w = max(abs(a - b) / b for a, b in zip(e1, e0))
            out.append('%s lists~dense err %.1e' % (dtype, w))
            bad += not (w < tol)
        print('graph %2d: n=%s rank=%s known=%.2f rel=%d theta=%d parts=%d  %s' % (g, list(n.values()), list(rank.values()), share, len(R),
                                                                                  len(Theta), parts, '  '.join(out)), flush=True)
    os.environ.pop('SKF_DFMC_SPARSE', None)
    os.environ.pop('SKF_KNOWN_PARTS', None)
    print('FAILED: %d' % bad if bad else 'all within tolerance')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 0))
