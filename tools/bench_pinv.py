"""Time skf_pinv_sym on the GPU: full-rank (Cholesky fast path) and rank-deficient (rank-revealing deflation;
SKF_PINV_JACOBI=1: the Jacobi eigen path)
Gram matrices of order n.    python tools/bench_pinv.py [n ...]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import scipy.linalg as spla
    import skfusion_amd._native as nat
    rt = nat.get_runtime()
    for n in [int(a) for a in sys.argv[1:]] or [64, 128, 256]:
        rs = np.random.RandomState(n)
        for name, G in (('full rank', rs.rand(4 * n, n)), ('rank n/2 ', rs.rand(n // 2, n))):
            A = np.ascontiguousarray(G.T @ G)
            a = rt.mem.from_host(A)
            k = rt.mem.empty(n * n * 8)
            nb = C.c_size_t()
            rt.call('skf_pinv_sym_workspace_bytes', n, C.byref(nb))
            ws = rt.mem.empty(nb.value)
            times = []
            for rep in range(6):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                rt.call('skf_pinv_sym', nat.SKF_F64, a.ptr, n, k.ptr, n, n, ws.ptr, nb.value, rt.mem.stream)
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
            dt = min(times[1:])
            reps = 40                                  # back to back: what an iteration pays (no host round trip per call)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                rt.call('skf_pinv_sym', nat.SKF_F64, a.ptr, n, k.ptr, n, n, ws.ptr, nb.value, rt.mem.stream)
            torch.cuda.synchronize()
            chained = (time.perf_counter() - t0) / reps
            if max(times[1:]) > 3 * dt:
                print('   (repetitions: %s ms)' % ' '.join('%.2f' % (t * 1e3) for t in times), flush=True)
            npad = (n + 1) // 2 * 2
            mat = (npad * npad * 8 + 255) // 256 * 256
            flags = rt.mem.to_host(ws, (nb.value // 4,), np.int32)[(3 * mat + (npad * 8 + 255) // 256 * 256) // 4 + 32]
            path = {1: 'inverse of the fast path', 2: 'deflation', 0: 'Jacobi eigen-solver'}.get(int(flags), '?')
            got = rt.mem.to_host(k, (n, n), np.float64)
            want = spla.pinv(A)
            err = np.linalg.norm(got - want) / np.linalg.norm(want)
            print('pinv n=%d %s: %.3f ms, %.3f ms back to back (%s), rel err vs scipy %.1e' % (n, name, dt * 1e3, chained * 1e3, path, err),
                  flush=True)


if __name__ == '__main__':
    main()
