"""Randomised small graphs through the device engine (small-graph schedule and general schedule) against the NumPy oracle:
    python tools/fuzz_small.py [n_graphs] [seed]
Sizes 1..300 objects, ranks 1..64, 2..4 types, random relation sets (incl. multi-relations), sparse constraints with empty
rows.  Prints the worst deviation per graph; exits non-zero above 1e-8 (f64) / 5e-3 (f32)."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import skfusion_amd._native as nat                                    # noqa: E402
from skfusion_amd._engine import DevicePlan, DeviceMatrix            # noqa: E402
from oracle import dfmf_oracle as orc                                 # noqa: E402
from helpers import relerr                                            # noqa: E402


def random_graph(rs):
    nt = rs.randint(2, 5)
    types = ['t%d' % k for k in range(nt)]
    n = {t: int(rs.choice([1, 2, 3, 5, 17, 63, 64, 65, 100, 129, 200, 257, 300])) for t in types}
    rank = {t: int(rs.choice([1, 2, 3, 7, 15, 16, 31, 32, 33, 50, 63, 64])) for t in types}
    R = {}
    pairs = [(a, b) for a in types for b in types if a != b]
    rs.shuffle(pairs)
    for (a, b) in pairs[:rs.randint(1, min(5, len(pairs)) + 1)]:
        if (b, a) in R:
            continue
        R[a, b] = [rs.rand(n[a], n[b]) for _ in range(1 + (rs.rand() < 0.25))]
    Theta = {}
    for t in types:
        if rs.rand() < 0.5 and n[t] >= 8:
            ms = []
            for _ in range(1 + (rs.rand() < 0.3)):
                T = np.where(rs.rand(n[t], n[t]) < 0.03, 0.05 * rs.randn(n[t], n[t]), 0.0)
                T = T + T.T
                T[rs.randint(n[t])] = 0.0
                ms.append(T)
            Theta[t, t] = ms
    G0 = {(t, t): rs.rand(n[t], rank[t]) + 0.05 for t in types}
    return types, n, rank, R, Theta, G0


def run(types, n, rank, R, Theta, G0, dtype, iters):
    rt = nat.get_runtime()
    npd = np.float64 if dtype == 'f64' else np.float32
    rel = [(i, j, m, None) for (i, j), ms in R.items() for m in ms]
    thetas = []
    for (t, _), ms in Theta.items():
        for T in ms:
            dm = DeviceMatrix(rt.mem.from_host(np.ascontiguousarray(T, dtype=npd)), T.shape)
            nnz = int(np.count_nonzero(T))
            dm.nnz = nnz if 0 < nnz <= T.shape[0] ** 2 // 16 else 0
            thetas.append((t, dm))
    plan = DevicePlan(types, n, rank, rel, thetas, nat.SKF_DFMF, dtype=dtype)
    for t in types:
        plan.set_factor(t, G0[t, t])
    plan.iterate(iters)
    G = {t: plan.get_factor(t) for t in types}
    S = [plan.get_backbone(k) for k in range(len(rel))]
    plan.close()
    return G, S


def main(n_graphs, seed, iters=3):
    rs = np.random.RandomState(seed)
    bad = 0
    for g in range(n_graphs):
        types, n, rank, R, Theta, G0 = random_graph(rs)
        Go, So = orc.dfmf(R, Theta, types, rank, max_iter=iters, G0=G0)
        so = [m for key in R for m in So[key]]
        line = []
        for dtype, tol in (('f64', 1e-8), ('f32', 5e-3)):
            for sched in ('small', 'general'):
                os.environ.pop('SKF_NO_SMALL_FUSED', None)
                if sched == 'general':
                    os.environ['SKF_NO_SMALL_FUSED'] = '1'
                G, S = run(types, n, rank, R, Theta, G0, dtype, iters)
                w = max([relerr(G[t], Go[t, t]) for t in types] + [relerr(a, b) for a, b in zip(S, so)])
                line.append('%s/%s %.1e' % (dtype, sched, w))
                if not (w < tol):
                    bad += 1
        os.environ.pop('SKF_NO_SMALL_FUSED', None)
        print('graph %2d: n=%s rank=%s rel=%d theta=%d  %s' % (g, list(n.values()), list(rank.values()), sum(len(v) for v in R.values()),
                                                              sum(len(v) for v in Theta.values()), '  '.join(line)), flush=True)
    print('FAILED: %d' % bad if bad else 'all within tolerance')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 0))
