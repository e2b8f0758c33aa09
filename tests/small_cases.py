"""A small graph that exercises every job kind of the four-launch schedule for small graphs (skf_small.h): object counts
that are no multiple of any tile (5 / 70 / 130 / 257), ranks 1 / 7 / 33 / 64, two relations between the same pair of types, a
type that is only ever a column type, negative relation values, several sparse constraints on one type and rows without
entries.  Shared by the emulator suite and the GPU suite."""
import numpy as np

import skfusion_amd._native as nat
from skfusion_amd._engine import DevicePlan, DeviceMatrix
from oracle import dfmf_oracle as orc
from helpers import relerr

TYPES = ['a', 'b', 'c', 'd']
N = {'a': 5, 'b': 70, 'c': 130, 'd': 257}
RANK = {'a': 1, 'b': 7, 'c': 33, 'd': 64}


def graph(seed=5):
    rs = np.random.RandomState(seed)
    R = {('a', 'b'): [rs.rand(5, 70)],
         ('b', 'c'): [rs.rand(70, 130), rs.rand(70, 130) - 0.3],     # a multi-relation; negative values
         ('b', 'd'): [rs.rand(70, 257)],
         ('c', 'd'): [rs.rand(130, 257) * (rs.rand(130, 257) < 0.3)]}
    T1 = np.where(rs.rand(130, 130) < 0.02, rs.randn(130, 130), 0.0)       # (CSR form: at most n^2 / 16 non-zeros)
    T1 = 0.1 * (T1 + T1.T)
    T2 = 0.25 * np.eye(130)
    T3 = np.where(rs.rand(257, 257) < 0.02, -0.02, 0.0)               # (heavier must-links make the objective unbounded)
    T3[11, :] = 0.0
    T3[200:, :] = 0.0
    Theta = {('c', 'c'): [T1, T2], ('d', 'd'): [T3]}
    G0 = {(t, t): rs.rand(N[t], RANK[t]) + 0.05 for t in TYPES}
    return R, Theta, G0


def run(dtype, iters=3):
    """(factors, backbones) of the device engine after `iters` iterations, constraints handed over with their counts."""
    R, Theta, G0 = graph()
    rt = nat.get_runtime()
    npd = np.float64 if dtype == 'f64' else np.float32
    rel = [(i, j, m, None) for (i, j), ms in R.items() for m in ms]
    thetas = []
    for (t, _), ms in Theta.items():
        for T in ms:
            dm = DeviceMatrix(rt.mem.from_host(np.ascontiguousarray(T, dtype=npd)), T.shape)
            dm.nnz = int(np.count_nonzero(T))
            thetas.append((t, dm))
    plan = DevicePlan(TYPES, N, RANK, rel, thetas, nat.SKF_DFMF, dtype=dtype)
    for t in TYPES:
        plan.set_factor(t, G0[t, t])
    plan.iterate(iters)
    G = {t: plan.get_factor(t) for t in TYPES}
    S = [plan.get_backbone(k) for k in range(len(rel))]
    plan.close()
    return G, S


def check(dtype, monkeypatch, iters=3):
    """fused schedule vs general schedule vs oracle; returns the worst deviations (fused vs oracle, fused vs general)."""
    R, Theta, G0 = graph()
    monkeypatch.delenv('SKF_NO_SMALL_FUSED', raising=False)
    Gf, Sf = run(dtype, iters)
    monkeypatch.setenv('SKF_NO_SMALL_FUSED', '1')
    Gs, Ss = run(dtype, iters)
    monkeypatch.delenv('SKF_NO_SMALL_FUSED', raising=False)
    Go, So = orc.dfmf(R, Theta, TYPES, RANK, max_iter=iters, G0=G0)
    so = [m for (i, j) in R for m in So[i, j]]
    worst_o = max([relerr(Gf[t], Go[t, t]) for t in TYPES] + [relerr(a, b) for a, b in zip(Sf, so)])
    worst_s = max([relerr(Gf[t], Gs[t]) for t in TYPES] + [relerr(a, b) for a, b in zip(Sf, Ss)])
    assert worst_s > 0.0, 'bit-identical results: the plan did not take the four-launch schedule (the sums differ in order)'
    # the register sweep with four pivots per barrier replays the one-pivot sweep operation by operation: the same bits
    monkeypatch.setenv('SKF_SMALL_SWEEP4', '0')
    G1, S1 = run(dtype, iters)
    monkeypatch.delenv('SKF_SMALL_SWEEP4', raising=False)
    for t in TYPES:
        np.testing.assert_array_equal(Gf[t], G1[t])
    for a, b in zip(Sf, S1):
        np.testing.assert_array_equal(a, b)
    return worst_o, worst_s
