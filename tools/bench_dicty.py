"""BASELINE config 2: dicty graph (ann 1219x116, expr 1219x282, Theta = ppi 1219x1219; ranks
50/15/5), Dfmf 100 iterations from the golden G0: GPU engines vs the CPU oracle (it/s)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    import torch
    from helpers import golden, dicty_graph, g0_from
    from oracle import dfmf_oracle as orc
    import skfusion_amd._native as nat
    from skfusion_amd._engine import DevicePlan, flatten_relations, flatten_thetas
    z = golden('c2_dicty.npz')
    R, Theta, types, rank = dicty_graph()
    G0 = g0_from(z, 'dfmf/', types)
    n = {'gene': 1219, 'go': 116, 'exc': 282}
    for dtype in ('f64', 'f32'):
        plan = DevicePlan(types, n, rank, flatten_relations(R), flatten_thetas(Theta), nat.SKF_DFMF, dtype=dtype)
        for t in types:
            plan.set_factor(t, G0[t, t])
        plan.iterate(5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan.iterate(100)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print('dicty %s: %.1f it/s (%.3f ms/iter)' % (dtype, 100 / dt, dt * 10), flush=True)
        plan.close()
    t0 = time.perf_counter()
    orc.dfmf(R, Theta, types, rank, max_iter=20, G0=G0)
    dt = time.perf_counter() - t0
    print('dicty CPU oracle (NumPy f64, %d cores): %.2f it/s (%.1f ms/iter)' % (os.cpu_count(), 20 / dt, dt * 50))


if __name__ == '__main__':
    main()
