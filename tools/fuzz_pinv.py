"""Randomised symmetric positive semi-definite matrices through the pseudo-inverse operator (skf_pinv_sym) against
scipy.linalg.pinv -- what `_dfmf.py:232` computes for every Gram matrix:
    python tools/fuzz_pinv.py [n_matrices] [seed] [--emulated]
Orders 2 .. 1023 (two thirds above 256: the multi-workgroup routes of round 6), ranks 1 .. n, Gram matrices of non-negative
factors with duplicated, zero and badly scaled latent columns.  A matrix whose non-zero spectrum reaches into the gap band of
the deflations (1e-10 .. 1e-7 of the largest eigenvalue) is compared at the looser bound of the eigen-solver route; a rank-deficient one at the bound of the
deflation tests (1e-8: a pivoted factorisation stops at its noise level); a full-rank one at eps * cond.  Prints
the route (1 = inverse written straight into K, 2 = one-workgroup deflation, 0 = eigen-solver) and the deviation per matrix;
exits non-zero above the bound."""
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import skfusion_amd._native as nat                                    # noqa: E402
from helpers import relerr                                            # noqa: E402
from test_emul_kernels import run_pinv                                # noqa: E402


def random_matrix(rs, small):
    if small:
        n = int(rs.choice([2, 3, 17, 33, 64, 65, 100, 129, 200, 257, 258, 300, 330]))
    elif rs.rand() < 0.67:
        n = int(rs.choice([257, 258, 288, 289, 300, 320, 321, 400, 448, 449, 512, 513, 600, 640, 777, 800, 1000, 1022, 1023]))
    else:
        n = int(rs.choice([2, 5, 31, 32, 33, 64, 65, 96, 128, 129, 200, 255, 256]))
    kind = rs.choice(['full', 'deficient', 'deficient', 'deficient', 'half', 'tiny'])
    rank = {'full': n, 'half': max(1, n // 2), 'tiny': int(rs.randint(1, 4))}.get(kind)
    if rank is None:
        rank = int(rs.randint(1, n + 1))
    rows = rank if rank < n else 4 * n + 3
    G = rs.rand(rows, n)
    what = [kind, 'rank %d' % min(rank, n)]
    if rs.rand() < 0.4 and n > 3:
        for _ in range(rs.randint(1, 4)):
            a, b = rs.choice(n, 2, replace=False)
            G[:, a] = G[:, b]
        what.append('duplicated')
    if rs.rand() < 0.3 and n > 3:
        G[:, rs.choice(n, rs.randint(1, 3), replace=False)] = 0.0
        what.append('zero columns')
    if rs.rand() < 0.3:
        G = G * 10.0 ** rs.uniform(-2, 2, size=n)
        what.append('scaled')
    A = G.T @ G
    return 0.5 * (A + A.T), ', '.join(what)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    count = int(args[0]) if args else 40
    seed = int(args[1]) if len(args) > 1 else 1
    emulated = '--emulated' in sys.argv
    import scipy.linalg as spla
    if emulated:
        from emul.runtime import emulated_runtime
        rt = emulated_runtime()
    else:
        rt = nat.get_runtime()
    rs = np.random.RandomState(seed)
    worst, bad, routes, t0 = 0.0, 0, {0: 0, 1: 0, 2: 0}, time.time()
    for k in range(count):
        A, what = random_matrix(rs, emulated)
        n = A.shape[0]
        w = np.linalg.eigvalsh(A)
        cut = max(w.max() * n * np.finfo(np.float64).eps, 0.0)
        kept = w[w > cut]
        band = bool(np.any((kept > 1e-10 * w.max()) & (kept < 1e-7 * w.max())))
        cond = kept.max() / kept.min() if kept.size else 1.0
        want = spla.pinv(A)
        route = []
        got = run_pinv(rt, nat.SKF_F64, A, route)
        dev = relerr(got, want)
        back = relerr(A @ got @ A, A)
        asym = np.abs(got - got.T).max() / max(np.abs(got).max(), 1e-300)
        # full rank: the sweep / Cholesky inverse, eps * cond; rank-deficient: a pivoted factorisation that stops at its noise
        # level leaves a residual of that size in A, i.e. (noise level) * cond in K -- the bound of the deflation tests
        deficient = kept.size < n
        bound = max(1e-8, 1e-14 * cond) if band else max(1e-8, 1e-12 * cond) if deficient else max(1e-11, 1e-15 * cond)
        ok = dev < bound and back < max(1e-10, 1e-14 * cond) and asym < 1e-12 and np.all(np.isfinite(got))
        routes[route[0]] = routes.get(route[0], 0) + 1
        worst = max(worst, dev / bound)
        bad += (not ok)
        print('%3d n=%4d %-44s cond %.1e%s route %d  dev %.2e (bound %.0e)  A K A %.1e  asym %.1e%s'
              % (k, n, what, cond, ' band' if band else '', route[0], dev, bound, back, asym, '' if ok else '   <-- FAIL'), flush=True)
    print('%d matrices in %.0f s, routes %s, worst deviation / bound %.3f, failures %d'
          % (count, time.time() - t0, routes, worst, bad))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
