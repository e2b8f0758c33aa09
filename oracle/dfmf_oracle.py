"""CPU oracle for the DFMF / DFMC / fold-in update loop  --  TEST INFRASTRUCTURE ONLY.

This module is a NumPy/SciPy *restatement* (float64, reference operation order) of the one
hot path of mims-harvard/scikit-fusion that this repository re-implements as HIP kernels.
It exists to *check* the HIP engine and to be timed as the CPU baseline (`bench.py`'s
``cpu_baseline`` leg, kind="port").  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s cpu_baseline leg may import it; the product package ``skfusion_amd`` never
does, and fails loudly if its HIP library is missing.

Parity pin: the functions below are checked against golden vectors produced by importing
the reference itself in the build container (``tools/gen_golden.py`` ->
``tests/golden/*.npz``; test: ``tests/test_oracle_golden.py``), i.e. parity is PINNED
against outputs of the reference run here.

Third-party arithmetic the reference delegates to (not under /root/reference):
``numpy.dot`` (OpenBLAS dgemm), ``numpy.nan_to_num``, ``scipy.linalg.pinv`` (SciPy >= 1.7
semantics: SVD via gesdd, singular values <= max(M,N)*eps*sigma_max dropped); versions in
the build container: numpy 2.2.6 / scipy 1.15.3 (the reference pins lower bounds only,
requirements.txt:1-3).

Every function cites the reference file:line it follows (paths relative to
/root/reference/skfusion/fusion/decomposition/).
"""
from collections import defaultdict

import numpy as np
import scipy.linalg as spla

EPS = np.finfo(float).eps      # reference: np.finfo(np.float).eps  (_dfmf.py:296)


# --------------------------------------------------------------------------------------
# initialisers  (_init.py:6-61)
# --------------------------------------------------------------------------------------
def _init_random(obj_types, n_obj, rank, R, rs):
    """_init.py:11-17 -- one ``rand(n_i, c_i)`` per type, in iteration order of obj_types."""
    G = {}
    for t in obj_types:
        G[t, t] = rs.rand(n_obj[t], rank[t])
    return G


def _init_random_c(obj_types, n_obj, rank, R, rs):
    """_init.py:20-41 -- columns drawn from the top floor(0.5*ncols) columns by L2 norm."""
    G = {}
    for t in obj_types:
        c = rank[t]
        G[t, t] = 1e-5 * np.ones((n_obj[t], c))
        for pair, R12 in R.items():
            if t not in pair:
                continue
            Rij = R12 if t == pair[0] else R12.T
            p_c = int(.2 * Rij.shape[1])
            l_c = int(.5 * Rij.shape[1])
            norms = [np.linalg.norm(Rij[:, k], 2) for k in range(Rij.shape[1])]
            # stable descending sort == sorted(enumerate(.), key=itemgetter(1), reverse=True)
            top = sorted(range(len(norms)), key=lambda k: norms[k], reverse=True)[:l_c]
            Gi = np.zeros(G[t, t].shape)
            for k in range(c):
                rs.shuffle(top)
                Gi[:, k] = Rij[:, top[:p_c]].mean(axis=1)
            G[t, t] += np.abs(Gi)
    return G


def _init_random_vcol(obj_types, n_obj, rank, R, rs):
    """_init.py:44-61 -- mean of p=floor(0.2*ncols) randomly chosen columns per factor column."""
    G = {}
    for t in obj_types:
        c = rank[t]
        G[t, t] = 1e-5 * np.ones((n_obj[t], c))
        for pair, R12 in R.items():
            if t not in pair:
                continue
            Rij = R12 if t == pair[0] else R12.T
            p_c = int(.2 * Rij.shape[1])
            Gi = np.zeros(G[t, t].shape)
            idx = np.arange(Rij.shape[1])
            for k in range(c):
                rs.shuffle(idx)
                Gi[:, k] = Rij[:, idx[:p_c]].mean(axis=1)
            G[t, t] += np.abs(Gi)
    return G


_INIT = {"random": _init_random, "random_c": _init_random_c, "random_vcol": _init_random_vcol}


def initialize(obj_types, n_obj, rank, R_first, init_type, random_state):
    """_init.py:6-8.  Unknown ``init_type`` -> KeyError, as in the reference."""
    return _INIT[init_type](obj_types, n_obj, rank, R_first, random_state)


def count_objects(obj_types, R):
    """_dfmf.py:95-124 (mismatches are only logged by the reference; here they raise)."""
    n = {}
    for (i, j), mats in R.items():
        for m in mats:
            for ax, t in enumerate((i, j)):
                if n.setdefault(t, m.shape[ax]) != m.shape[ax]:
                    raise ValueError("relation (%s,%s) dimension mismatch" % (i, j))
    return n


# --------------------------------------------------------------------------------------
# shared pieces of one iteration
# --------------------------------------------------------------------------------------
def _split(x):
    """_dfmf.py:256-258: t = x > 0; xp = t*x; xn = (t-1)*x  (xn >= 0)."""
    t = x > 0
    return np.multiply(t, x), np.multiply(t - 1, x)


def _theta_split(Theta):
    """_dfmf.py:203-208."""
    Tp, Tn = defaultdict(list), defaultdict(list)
    for r, thetas in Theta.items():
        for th in thetas:
            p, n = _split(np.asarray(th, dtype=float))
            Tp[r].append(p)
            Tn[r].append(n)
    return Tp, Tn


def _update_S(R, G, nan_to_num=True):
    """_dfmf.py:228-239 (same in _dfmc.py:297-315): K_i = pinv(nan_to_num(G_i^T G_i));
    S_ij^l = K_i (G_i^T (R_ij^l (G_j K_j))), evaluated right-to-left with nan_to_num after
    every block product (__bdot, _dfmf.py:19-41).  Returns (S, K)."""
    K, GK = {}, {}
    for r, Gr in G.items():
        K[r] = spla.pinv(np.nan_to_num(np.dot(Gr.T, Gr)))
        GK[r] = np.nan_to_num(np.dot(Gr, K[r]))
    S = {}
    for (i, j), mats in R.items():
        S[i, j] = []
        for Rl in mats:
            t2 = np.nan_to_num(np.dot(Rl, GK[j, j]))
            t3 = np.nan_to_num(np.dot(G[i, i].T, t2))
            S[i, j].append(np.nan_to_num(np.dot(K[i, i], t3)))
    return S, K


def _relation_terms(Rl, Gi, Gj, Sl, nan_to_num):
    """_dfmf.py:254-276 / _dfmc.py:152-170 (_update_G_for_Rij has no nan_to_num)."""
    f = np.nan_to_num if nan_to_num else (lambda x: x)
    t1 = f(np.dot(Rl, np.dot(Gj, Sl.T)))
    t1p, t1n = _split(t1)
    t2 = f(np.dot(Sl, np.dot(Gj.T, np.dot(Gj, Sl.T))))
    t2p, t2n = _split(t2)
    t4 = f(np.dot(Rl.T, np.dot(Gi, Sl)))
    t4p, t4n = _split(t4)
    t5 = f(np.dot(Sl.T, np.dot(Gi.T, np.dot(Gi, Sl))))
    t5p, t5n = _split(t5)
    Ei = t1p + np.dot(Gi, t2n)
    Di = t1n + np.dot(Gi, t2p)
    Ej = t4p + np.dot(Gj, t5n)
    Dj = t4n + np.dot(Gj, t5p)
    return (Ei, Di), (Ej, Dj)


def _update_G(R, G, S, Tp, Tn, nan_to_num):
    """_dfmf.py:246-296 (Jacobi style: every G is replaced from the OLD factors)."""
    E = {r: np.zeros(Gr.shape) for r, Gr in G.items()}
    D = {r: np.zeros(Gr.shape) for r, Gr in G.items()}
    for (i, j), mats in R.items():
        for l, Rl in enumerate(mats):
            (Ei, Di), (Ej, Dj) = _relation_terms(Rl, G[i, i], G[j, j], S[i, j][l], nan_to_num)
            E[i, i] += Ei
            D[i, i] += Di
            E[j, j] += Ej
            D[j, j] += Dj
    for r, ths in Tp.items():                         # _dfmf.py:285-288
        for th in ths:
            D[r] += np.dot(th, G[r])
    for r, ths in Tn.items():                         # _dfmf.py:289-292
        for th in ths:
            E[r] += np.dot(th, G[r])
    newG = {}
    for r in G:                                       # _dfmf.py:294-296
        newG[r] = np.multiply(G[r], np.sqrt(np.divide(E[r], np.maximum(D[r], EPS))))
    return newG


def relation_errors(R, G, S):
    """_dfmf.py:306-316: per-relation Frobenius error ||R - G_i S G_j^T||_F."""
    errs = {}
    for (i, j), mats in R.items():
        errs[i, j] = [np.linalg.norm(Rl - np.dot(G[i, i], np.dot(S[i, j][l], G[j, j].T)), "fro")
                      for l, Rl in enumerate(mats)]
    return errs


def relation_errors_blocked(R, G, S, rows=4096):
    """The same errors (_dfmf.py:306-316) summed over row blocks of the relation: the n_i x n_j reconstruction is never
    held as a whole (config 3 at full size: 40 GB per relation).  Equal to relation_errors up to the order of the sum."""
    errs = {}
    for (i, j), mats in R.items():
        errs[i, j] = []
        for l, Rl in enumerate(mats):
            H = np.dot(S[i, j][l], G[j, j].T)
            sq = 0.0
            for r0 in range(0, Rl.shape[0], rows):
                d = Rl[r0:r0 + rows] - np.dot(G[i, i][r0:r0 + rows], H)
                sq += float(np.vdot(d, d))
            errs[i, j].append(np.sqrt(sq))
    return errs


# --------------------------------------------------------------------------------------
# the three solver entry points
# --------------------------------------------------------------------------------------
def _system_error(R, G, S):
    """_dfmf.py:306-316 / _dfmc.py:376-386: the UNSQUARED sum of the per-relation Frobenius errors."""
    e = relation_errors(R, G, S)
    return sum(sum(v) for v in e.values())


def dfmf(R, Theta, obj_types, obj_type2rank, max_iter=10, init_type="random_vcol",
         callback=None, random_state=None, G0=None, stopping=None, stopping_system=None, compute_err=False):
    """Restatement of ``dfmf()`` (_dfmf.py:127-327).  ``G0`` (dict keyed (t,t)) overrides
    the initialiser so that parity tests do not depend on set-iteration order.
    stopping = ((row, col), eps) / stopping_system = eps: the early-stopping rules of _dfmf.py:213-221 -- checked from the
    third iteration on, on the change of the target relation's error (:301-304; the reference's expression subtracts the
    LISTS of a pair and only means something for the first relation of the pair: that one is taken) / of the summed
    objective (:306-319; forces compute_err, :198-200)."""
    R = {k: [np.asarray(m, dtype=float) for m in v] for k, v in R.items()}
    n_obj = count_objects(obj_types, R)
    if G0 is None:
        R_first = {k: v[0] for k, v in R.items()}     # _dfmf.py:191
        G = initialize(obj_types, n_obj, obj_type2rank, R_first, init_type, random_state)
    else:
        G = {k: np.array(v, dtype=float) for k, v in G0.items()}
    Tp, Tn = _theta_split(Theta)
    S = None
    if stopping_system:
        compute_err = True                            # _dfmf.py:198-200
    err, err_system = (None, None), (None, None)
    for it in range(max_iter):
        if it > 1 and stopping and err[1] - err[0] < stopping[1]:                            # _dfmf.py:213-216
            break
        if it > 1 and stopping_system and err_system[1] - err_system[0] < stopping_system:   # :217-221
            break
        S, _ = _update_S(R, G)
        G = _update_G(R, G, S, Tp, Tn, nan_to_num=True)
        if stopping:                                  # :301-304
            i, j = stopping[0]
            err = (np.linalg.norm(R[i, j][0] - np.dot(G[i, i], np.dot(S[i, j][0], G[j, j].T))), err[0])
        if compute_err:                               # :306-319
            err_system = (_system_error(R, G, S), err_system[0])
        if callback:
            callback(G, S, it)
    return G, S


def dfmc(R, M, Theta, obj_types, obj_type2rank, max_iter=10, init_type="random_vcol",
         callback=None, random_state=None, G0=None, stopping=None, stopping_system=None, compute_err=False):
    """Restatement of ``dfmc()`` (_dfmc.py:181-397): dfmf + completion of masked entries
    (zeroed at iteration 0, :287-292; overwritten with G_i S G_j^T after every S update,
    :319-325).  The relation data is copied (:268), inputs are never mutated.
    stopping = (((row, col), l), eps) (_dfmc.py:370-374) / stopping_system = eps (:376-389), both on the WORKING copy of
    the relations (completed entries included), checked from the third iteration on (:271-279)."""
    R = {k: [np.array(m, dtype=float) for m in v] for k, v in R.items()}      # copies
    n_obj = count_objects(obj_types, R)
    if G0 is None:
        R_first = {k: v[0] for k, v in R.items()}
        G = initialize(obj_types, n_obj, obj_type2rank, R_first, init_type, random_state)
    else:
        G = {k: np.array(v, dtype=float) for k, v in G0.items()}
    Tp, Tn = _theta_split(Theta)
    S = None
    if stopping_system:
        compute_err = True                            # _dfmc.py:256-258
    err, err_system = (None, None), (None, None)
    for it in range(max_iter):
        if it > 1 and stopping and err[1] - err[0] < stopping[1]:                            # _dfmc.py:271-274
            break
        if it > 1 and stopping_system and err_system[1] - err_system[0] < stopping_system:   # :275-279
            break
        if it == 0:
            for r in M:
                for l in range(len(R[r])):
                    if M[r][l] is not None:
                        R[r][l][M[r][l]] = 0.
        S, _ = _update_S(R, G)
        for r in M:
            for l in range(len(M[r])):
                if M[r][l] is None:
                    continue
                i, j = r
                app = np.dot(G[i, i], np.dot(S[i, j][l], G[j, j].T))
                R[r][l][M[r][l]] = app[M[r][l]]
        G = _update_G(R, G, S, Tp, Tn, nan_to_num=False)
        if stopping:                                  # _dfmc.py:370-374
            (i, j), l = stopping[0]
            err = (np.linalg.norm(R[i, j][l] - np.dot(G[i, i], np.dot(S[i, j][l], G[j, j].T))), err[0])
        if compute_err:                               # :376-389
            err_system = (_system_error(R, G, S), err_system[0])
        if callback:
            callback(G, S, it)
    return G, S


def transform(R_ij, Theta_i, target, obj_type2rank, G, S, max_iter=10, init_type="random_c",
              callback=None, random_state=None, G0=None):
    """Restatement of ``transform()`` (_dfmf.py:330-458): only G_target moves; S and every
    other G are frozen; no pinv, no nan_to_num; callback(G_i, iter)."""
    R_ij = {k: [np.asarray(m, dtype=float) for m in v] for k, v in R_ij.items()}
    if G0 is None:
        if not isinstance(random_state, np.random.RandomState):
            random_state = np.random.RandomState(random_state)
        n_t = [R_ij[i, j][0].shape[0 if target == i else 1] for i, j in R_ij]
        R_first = {k: v[0] for k, v in R_ij.items()}
        Gi = initialize([target], {target: n_t[0]}, obj_type2rank, R_first,
                        init_type, random_state)[target, target]
    else:
        Gi = np.array(G0, dtype=float)
    Tp, Tn = [], []
    for r, ths in Theta_i.items():                    # _dfmf.py:357-363
        for th in ths:
            p, n = _split(np.asarray(th, dtype=float))
            Tp.append(p)
            Tn.append(n)
    for it in range(max_iter):
        E = np.zeros(Gi.shape)
        D = np.zeros(Gi.shape)
        for (i, j), mats in R_ij.items():
            for l, Rl in enumerate(mats):
                Sl = S[i, j][l]
                if i is target:                       # _dfmf.py:392-405
                    Gj = G[j, j]
                    t1p, t1n = _split(np.dot(Rl, np.dot(Gj, Sl.T)))
                    t2p, t2n = _split(np.dot(Sl, np.dot(Gj.T, np.dot(Gj, Sl.T))))
                    E += t1p + np.dot(Gi, t2n)
                    D += t1n + np.dot(Gi, t2p)
                if j is target:                       # _dfmf.py:407-419
                    Gr = G[i, i]
                    t4p, t4n = _split(np.dot(Rl.T, np.dot(Gr, Sl)))
                    t5p, t5n = _split(np.dot(Sl.T, np.dot(Gr.T, np.dot(Gr, Sl))))
                    E += t4p + np.dot(Gi, t5n)
                    D += t4n + np.dot(Gi, t5p)
        for th in Tp:                                 # _dfmf.py:421-425
            D += np.dot(th, Gi)
        for th in Tn:
            E += np.dot(th, Gi)
        Gi = np.multiply(Gi, np.sqrt(np.divide(E, np.maximum(D, EPS))))
        if callback:
            callback(Gi, it)
    return Gi


# --------------------------------------------------------------------------------------
# the algebraically identical 2-GEMM form the HIP engine implements
# --------------------------------------------------------------------------------------
def dfmf_two_gemm_step(R, G, Tp, Tn, M=None, nan_to_num=True, dtype=np.float64):
    """One iteration in the engine's schedule: P = R G_j and Q = R^T G_i are the only
    products that touch R; S = K_i (G_i^T P) K_j; A = P S^T; C = Q S.  Kept here as the
    executable spec of the kernel schedule (tests compare it with `dfmf`/`dfmc` above)."""
    f = np.nan_to_num if nan_to_num else (lambda x: x)
    GtG = {r: f(np.dot(Gr.T, Gr)) for r, Gr in G.items()}
    K = {r: spla.pinv(GtG[r].astype(np.float64)).astype(dtype) for r in G}
    E = {r: np.zeros(Gr.shape, dtype) for r, Gr in G.items()}
    D = {r: np.zeros(Gr.shape, dtype) for r, Gr in G.items()}
    S = {}
    for (i, j), mats in R.items():
        S[i, j] = []
        for l, Rl in enumerate(mats):
            Gi, Gj = G[i, i], G[j, j]
            P = np.dot(Rl, Gj)
            Sl = f(np.dot(K[i, i], np.dot(np.dot(Gi.T, P), K[j, j])))
            S[i, j].append(Sl)
            if M is not None and M.get((i, j)) is not None and M[i, j][l] is not None:
                m = M[i, j][l]
                Rl[m] = np.dot(Gi, np.dot(Sl, Gj.T))[m]
                P = np.dot(Rl, Gj)
            Q = np.dot(Rl.T, Gi)
            Ap, An = _split(f(np.dot(P, Sl.T)))
            Bp, Bn = _split(f(np.dot(Sl, np.dot(GtG[j, j], Sl.T))))
            Cp, Cn = _split(f(np.dot(Q, Sl)))
            Dp, Dn = _split(f(np.dot(Sl.T, np.dot(GtG[i, i], Sl))))
            E[i, i] += Ap + np.dot(Gi, Bn)
            D[i, i] += An + np.dot(Gi, Bp)
            E[j, j] += Cp + np.dot(Gj, Dn)
            D[j, j] += Cn + np.dot(Gj, Dp)
    for r, ths in Tp.items():
        for th in ths:
            D[r] += np.dot(th, G[r])
    for r, ths in Tn.items():
        for th in ths:
            E[r] += np.dot(th, G[r])
    eps = np.asarray(EPS, dtype)
    newG = {r: (G[r] * np.sqrt(E[r] / np.maximum(D[r], eps))).astype(dtype) for r in G}
    return newG, S


def rmse(R, Gi, S, Gj):
    """examples/movielens_completion.py:89-90 style RMSE of one relation."""
    diff = R - np.dot(Gi, np.dot(S, Gj.T))
    return float(np.sqrt(np.mean(diff * diff)))


# --------------------------------------------------------------------------------------
# counter-based synthetic data (shared, bit for bit, with the HIP fill kernel
# `skf_fill_uniform`, csrc/skf_kernels.h): value(seed, idx) = top 24 bits of
# splitmix64(seed * GOLDEN + idx) scaled to [0,1).  Not part of the reference.
# --------------------------------------------------------------------------------------
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def hash_uniform(seed, start, count):
    with np.errstate(over='ignore'):
        z = np.uint64(seed) * _GOLDEN + np.arange(start, start + count, dtype=np.uint64)
        z = z + _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)


def hash_uniform_at(seed, index):
    """The same generator at arbitrary linear element indices (rows / columns of a matrix too large to build)."""
    with np.errstate(over='ignore'):
        z = np.uint64(seed) * _GOLDEN + np.asarray(index, dtype=np.uint64)
        z = z + _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)


def hash_uniform_matrix(seed, n_rows, n_cols):
    return hash_uniform(seed, 0, n_rows * n_cols).reshape(n_rows, n_cols)
