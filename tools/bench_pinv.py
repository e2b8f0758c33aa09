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
            for rep in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                rt.call('skf_pinv_sym', nat.SKF_F64, a.ptr, n, k.ptr, n, n, ws.ptr, nb.value, rt.mem.stream)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            got = rt.mem.to_host(k, (n, n), np.float64)
            want = spla.pinv(A)
            err = np.linalg.norm(got - want) / np.linalg.norm(want)
            print('pinv n=%d %s: %.3f ms, rel err vs scipy %.1e' % (n, name, dt * 1e3, err), flush=True)


if __name__ == '__main__':
    main()
