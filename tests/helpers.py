"""Shared builders for the parity tests: the graphs behind tests/golden/*.npz."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TYPES = ['t1', 't2', 't3']


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def readme_graph():
    """reference README.md:50-64 (BASELINE config 1)."""
    R = {('t1', 't2'): [np.random.RandomState(0).rand(50, 100)],
         ('t1', 't3'): [np.random.RandomState(1).rand(50, 40)],
         ('t2', 't3'): [np.random.RandomState(2).rand(100, 40)]}
    return R, list(TYPES), {'t1': 10, 't2': 20, 't3': 30}


def probe_graph(z):
    """multi-relation / negative relation / Theta / masks graph stored in probe_multirel.npz."""
    R = {('t1', 't2'): [z['R_t1_t2_0'], z['R_t1_t2_1']],
         ('t1', 't3'): [z['R_t1_t3_0']],
         ('t2', 't3'): [z['R_t2_t3_0']]}
    Theta = {('t1', 't1'): [z['Theta_t1_0']], ('t2', 't2'): [z['Theta_t2_0'], z['Theta_t2_1']]}
    M = {('t1', 't2'): [z['M_t1_t2_0'], None], ('t1', 't3'): [None], ('t2', 't3'): [z['M_t2_t3_0']]}
    return R, Theta, M, list(TYPES), {'t1': 6, 't2': 5, 't3': 4}


def rank_deficient_graph(z):
    R = {('t1', 't2'): [z['R_t1_t2_0']], ('t1', 't3'): [z['R_t1_t3_0']]}
    return R, list(TYPES), {'t1': 50, 't2': 30, 't3': 10}


def dicty_graph():
    """BASELINE config 2 inputs (values of reference datasets/base.py:47-61, as a data fixture)."""
    z = golden('dicty_inputs.npz')
    shp = tuple(z['ann_shape'])
    ann = np.unpackbits(z['ann_bits'], axis=1)[:, :shp[1]].astype(np.float64)
    x = z['expr_milli'].astype(np.int64) / 1000.0
    expr = np.log(np.maximum(x, np.finfo(float).eps))          # datasets/base.py:57
    ppi = np.zeros(tuple(z['ppi_shape']))
    ppi[z['ppi_rows'], z['ppi_cols']] = z['ppi_vals']
    R = {('gene', 'go'): [ann], ('gene', 'exc'): [expr]}
    Theta = {('gene', 'gene'): [ppi]}
    return R, Theta, ['gene', 'go', 'exc'], {'gene': 50, 'go': 15, 'exc': 5}


def c3_scaled_graph(z):
    from oracle.dfmf_oracle import hash_uniform_matrix
    n1, n2, n3 = [int(v) for v in z['shape']]
    c = [int(v) for v in z['ranks']]
    ds, gs = z['data_seeds'], z['g0_seeds']
    R = {('t1', 't2'): [hash_uniform_matrix(int(ds[0]), n1, n2)],
         ('t1', 't3'): [hash_uniform_matrix(int(ds[1]), n1, n3)],
         ('t2', 't3'): [hash_uniform_matrix(int(ds[2]), n2, n3)]}
    G0 = {('t1', 't1'): hash_uniform_matrix(int(gs[0]), n1, c[0]),
          ('t2', 't2'): hash_uniform_matrix(int(gs[1]), n2, c[1]),
          ('t3', 't3'): hash_uniform_matrix(int(gs[2]), n3, c[2])}
    return R, G0, list(TYPES), dict(zip(TYPES, c))


def g0_from(z, prefix, types):
    return {(t, t): z['%sG0_%s' % (prefix, t)] for t in types}


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


class Snapshots:
    """callback(G, S, it) recorder mirroring the golden capture."""

    def __init__(self, keep):
        self.keep, self.snap = set(keep), {}

    def __call__(self, G, S, it):
        if it in self.keep:
            self.snap[it] = ({r: np.array(v) for r, v in G.items()},
                             {r: [np.array(s) for s in v] for r, v in S.items()})


def compare_snapshots(z, prefix, snaps, tol, rows=None):
    """Compare recorded (G,S) with every `prefix`G_*_itN / S_*_itN array in the golden file."""
    worst = 0.0
    n = 0
    for key in z.files:
        if not key.startswith(prefix):
            continue
        name = key[len(prefix):]
        if name.startswith('G_') and '_it' in name:
            t, it = name[2:].rsplit('_it', 1)
            if int(it) not in snaps:
                continue
            got = snaps[int(it)][0][t, t]
            want = z[key]
            got = got[:want.shape[0]]
        elif name.startswith('S_') and '_it' in name:
            body, it = name[2:].rsplit('_it', 1)
            if int(it) not in snaps:
                continue
            i, j, l = body.rsplit('_', 2)
            got = snaps[int(it)][1][i, j][int(l)]
            want = z[key]
        else:
            continue
        e = relerr(got, want)
        worst = max(worst, e)
        n += 1
        assert e <= tol, '%s: rel err %.3e > %.1e' % (key, e, tol)
    assert n > 0, 'no golden arrays matched prefix %r' % prefix
    return worst
