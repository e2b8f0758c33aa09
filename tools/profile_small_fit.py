"""Where the host time of `Dfmf(n_run=10, max_iter=100).fuse` on the README graph goes (cProfile, cumulative)."""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from skfusion_amd import fusion
    t1, t2, t3 = fusion.ObjectType('Type 1', 10), fusion.ObjectType('Type 2', 20), fusion.ObjectType('Type 3', 30)
    R12, R13, R23 = (np.random.RandomState(s).rand(*shp) for s, shp in ((0, (50, 100)), (1, (50, 40)), (2, (100, 40))))
    graph = fusion.FusionGraph([fusion.Relation(R12, t1, t2), fusion.Relation(R13, t1, t3), fusion.Relation(R23, t2, t3)])
    for n_run in (1, 10):
        fusion.Dfmf(max_iter=5, n_run=n_run, random_state=0).fuse(graph)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            fusion.Dfmf(max_iter=100, n_run=n_run, random_state=0).fuse(graph)
            best = min(best, time.perf_counter() - t0)
        print('Dfmf(max_iter=100, n_run=%d).fuse: best of 5 %.2f ms' % (n_run, best * 1e3))
        pr = cProfile.Profile()
        pr.enable()
        fusion.Dfmf(max_iter=100, n_run=n_run, random_state=0).fuse(graph)
        pr.disable()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28)
        print(s.getvalue()[:5000])


if __name__ == '__main__':
    main()
