"""Wall time of the class-level API on the reference's README graph (50x100 / 50x40 / 100x40, ranks
10/20/30): Dfmf(max_iter=100).fuse(graph), 5 runs after a warm-up, vs the NumPy oracle."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from skfusion_amd import fusion
    from oracle import dfmf_oracle as orc
    t1, t2, t3 = fusion.ObjectType('Type 1', 10), fusion.ObjectType('Type 2', 20), fusion.ObjectType('Type 3', 30)
    R12, R13, R23 = (np.random.RandomState(s).rand(*shp) for s, shp in ((0, (50, 100)), (1, (50, 40)), (2, (100, 40))))
    graph = fusion.FusionGraph([fusion.Relation(R12, t1, t2), fusion.Relation(R13, t1, t3), fusion.Relation(R23, t2, t3)])
    for dtype in ('f64', 'f32'):
        fusion.Dfmf(max_iter=5, random_state=0, dtype=dtype).fuse(graph)                 # warm-up
        t0 = time.perf_counter()
        for k in range(5):
            fusion.Dfmf(max_iter=100, random_state=k, dtype=dtype).fuse(graph)
        dt = (time.perf_counter() - t0) / 5
        print('README graph, Dfmf(max_iter=100, dtype=%s).fuse: %.1f ms per fit (%.3f ms per iteration incl. host set-up)'
              % (dtype, dt * 1e3, dt * 10))
    for n_jobs in (1, 2, 5, 10):
        fusion.Dfmf(max_iter=5, n_run=10, n_jobs=n_jobs, random_state=0).fuse(graph)     # warm-up
        t0 = time.perf_counter()
        fusion.Dfmf(max_iter=100, n_run=10, n_jobs=n_jobs, random_state=0).fuse(graph)
        dt = time.perf_counter() - t0
        print('README graph, Dfmf(max_iter=100, n_run=10, n_jobs=%d).fuse: %.1f ms (%.1f ms per restart)'
              % (n_jobs, dt * 1e3, dt * 100))
    R = {('a', 'b'): [R12], ('a', 'c'): [R13], ('b', 'c'): [R23]}
    t0 = time.perf_counter()
    orc.dfmf(R, {}, ['a', 'b', 'c'], {'a': 10, 'b': 20, 'c': 30}, max_iter=100, init_type='random_c',
             random_state=np.random.RandomState(0))
    print('NumPy oracle, same fit: %.1f ms' % ((time.perf_counter() - t0) * 1e3))


if __name__ == '__main__':
    main()
