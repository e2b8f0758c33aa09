"""Randomised graphs through the ownership-sharded fit (SKF_OPT_OWNED_ROWS, 2..4 ranks driven by threads of this process,
tests/helpers.fit_owned) against the NumPy oracle:
    python tools/fuzz_owned.py [n_graphs] [seed]
2..4 types of 1..700 objects (fewer objects than ranks included), ranks on both sides of 64 (one stream / three streams),
random relation sets incl. multi-relations and types that sit on one side only, DFMF and DFMC with masks of every density
(all known, None, 2 % known: the list form under row ownership), sparse and dense constraints.
Every engine against the oracle, next to the single-device fit of the same engine against the oracle: the sharded fit must
be within 1e-8 (f64) / 2e-4 (f32) / 3e-2 (bf16) OR within 10 x the single-device deviation OR (f32) within 1e8 x the f64
deviation of the same sharded fit -- graphs with (nearly)
rank-deficient Gram matrices (fewer objects than latent dimensions, square factors) amplify every rounding difference, a
single-device fit as much as a sharded one (64 objects at rank 64: 2e-7 in f64, 1e-2 in f32 on one device); the bf16 engine
is only run where every type has at least twice as many objects as latent dimensions.
Prints the deviations per graph; exits non-zero above the tolerances."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import dfmf_oracle as orc                                                 # noqa: E402
from helpers import relerr, fit_owned                                                 # noqa: E402
from skfusion_amd.fusion.decomposition import _dfmf, _dfmc                            # noqa: E402


def random_graph(rs, wide):
    nt = rs.randint(2, 5)
    types = ['t%d' % k for k in range(nt)]
    sizes = [1, 2, 3, 5, 17, 63, 64, 65, 100, 129, 200, 257, 300] + ([400, 513, 700] if wide else [])
    if wide and rs.rand() < 0.6:                  # every type at least twice as large as its rank: the bf16 engine's domain
        sizes = [300, 400, 513, 700]
    n = {t: int(rs.choice(sizes)) for t in types}
    if wide:                                      # at least one rank above 64: the three-stream schedule; small ones beside it
        rank = {t: int(rs.choice([3, 16, 50, 64, 65, 72, 96, 130])) for t in types}
        rank[types[rs.randint(nt)]] = int(rs.choice([65, 72, 96, 130]))
    else:
        rank = {t: int(rs.choice([1, 2, 3, 7, 15, 16, 31, 32, 33, 50, 63, 64])) for t in types}
    R, M = {}, {}
    pairs = [(a, b) for a in types for b in types if a != b]
    rs.shuffle(pairs)
    for (a, b) in pairs[:rs.randint(1, min(5, len(pairs)) + 1)]:
        if (b, a) in R:
            continue
        mats, masks = [], []
        for _ in range(1 + (rs.rand() < 0.25)):
            kind = rs.randint(4)
            if kind == 0:
                mats.append((rs.rand(n[a], n[b]) < 0.2).astype(np.float64))            # a 0/1 relation
            else:
                mats.append(rs.rand(n[a], n[b]) - (0.2 if kind == 1 else 0.0))
            mk = rs.randint(4)
            masks.append(None if mk == 0 else rs.rand(n[a], n[b]) < (0.02 if mk == 1 else 0.6 if mk == 2 else 1.0))
        R[a, b], M[a, b] = mats, masks
    used = {t for key in R for t in key}
    types = [t for t in types if t in used]       # (count_objects needs every type in a relation)
    Theta = {}
    for t in types:
        if rs.rand() < 0.5 and n[t] >= 8:
            ms = []
            for _ in range(1 + (rs.rand() < 0.3)):
                dens = 0.03 if rs.rand() < 0.7 else 0.5                                # sparse (CSR) / dense product
                T = np.where(rs.rand(n[t], n[t]) < dens, 0.02 * rs.randn(n[t], n[t]), 0.0)
                T = T + T.T + 0.05 * np.eye(n[t])
                T[rs.randint(n[t])] = 0.0
                T = 0.5 * (T + T.T)
                ms.append(T)
            Theta[t, t] = ms
    G0 = {(t, t): rs.rand(n[t], rank[t]) + 0.05 for t in types}
    return types, {t: n[t] for t in types}, {t: rank[t] for t in types}, R, M, Theta, G0


def worst(out, Gs, Ss, types):
    w = 0.0
    for G, S in out:
        for t in types:
            w = max(w, relerr(G[t, t], Gs[t, t]))
        for key in Ss:
            for a, b in zip(S[key], Ss[key]):
                w = max(w, relerr(a, b))
    return w


def main(n_graphs, seed, iters=3):
    if os.environ.get('SKF_FUZZ_EMUL'):           # the host emulator instead of the GPU (slow: keep n_graphs small)
        from emul.runtime import emulated_runtime, use_runtime
        with use_runtime(emulated_runtime()):
            return fuzz(n_graphs, seed, iters)
    return fuzz(n_graphs, seed, iters)


def fuzz(n_graphs, seed, iters):
    rs = np.random.RandomState(seed)
    bad = 0
    for g in range(n_graphs):
        wide = bool(g % 2)
        types, n, rank, R, M, Theta, G0 = random_graph(rs, wide)
        size = int(rs.randint(2, 5))
        line = []
        for variant in ('dfmf', 'dfmc'):
            if variant == 'dfmf':
                Go, So = orc.dfmf(R, Theta, types, rank, max_iter=iters, G0=G0)
            else:
                Go, So = orc.dfmc(R, M, Theta, types, rank, max_iter=iters, G0=G0)
            well = all(n[t] >= 2 * rank[t] for t in types)
            dtypes = [('f64', 1e-8), ('f32', 2e-4)] + ([('bf16', 3e-2)] if wide and well and min(rank.values()) >= 16 else [])
            w64 = 0.0
            for dtype, tol in dtypes:
                if variant == 'dfmf':
                    Gs, Ss = _dfmf.dfmf(R, Theta, types, rank, max_iter=iters, G0=G0, dtype=dtype)
                else:
                    Gs, Ss = _dfmc.dfmc(R, M, Theta, types, rank, max_iter=iters, G0=G0, dtype=dtype)
                one = worst([(Gs, Ss)], Go, So, types)
                try:
                    out, _, _ = fit_owned(variant, R, M, Theta, types, rank, G0, iters, size, dtype=dtype)
                    w = worst(out, Go, So, types)
                except Exception as e:                                                  # noqa: BLE001
                    w = float('inf')
                    line.append('%s/%s %s: %s' % (variant, dtype, type(e).__name__, str(e)[:120]))
                line.append('%s/%s %.1e (one device %.1e)' % (variant, dtype, w, one))
                if dtype == 'f64':
                    w64 = w
                # f32: the same graph's f64 deviation measures how far its conditioning amplifies a rounding error (1e-13 on
                # a well-conditioned graph); 1e8 = a fifth of eps(f32) / eps(f64) carries that over
                if not (w < max(tol, 10.0 * one, 1e8 * w64 if dtype == 'f32' else 0.0)):
                    bad += 1
                    line.append('<-- above %.0e' % tol)
        print('graph %2d: W=%d n=%s rank=%s rel=%d theta=%d  %s' % (g, size, list(n.values()), list(rank.values()),
                                                                   sum(len(v) for v in R.values()), sum(len(v) for v in Theta.values()),
                                                                   '  '.join(line)), flush=True)
    print('FAILED: %d' % bad if bad else 'all within tolerance')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 12, int(sys.argv[2]) if len(sys.argv) > 2 else 0))
