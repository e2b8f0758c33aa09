"""Stand-alone timing of the WIDE products of an iteration -- Gram = G^T G and W = G_i^T P (f32 operands, f64 accumulation,
split over K with a fixed-order reduce) -- at BASELINE config-3 shapes, over the number of K slices.
    python tools/bench_wide.py            # SKF_GRAM_SYM=0: Gram products on every tile"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import skfusion_amd._native as nat
    from skfusion_amd._engine import fill_uniform
    rt = nat.get_runtime()
    ws = rt.mem.empty(512 << 20)
    shapes = [('Gram t1', 128, 128, 50000, True), ('Gram t2', 256, 256, 100000, True), ('Gram t3', 256, 256, 40000, True),
              ('W12 = G1^T P12', 128, 256, 50000, False), ('W23 = Q23^T G3', 256, 256, 40000, False)]
    for label, M, N, K, gram in shapes:
        A = fill_uniform((K, M), 1, 'f32')
        B = A if gram else fill_uniform((K, N), 2, 'f32')
        Cm = rt.mem.empty(M * N * 8)
        for splits in (0, 16, 32, 48, 64, 85, 96, 128, 170, 192, 256):
            d = nat.GemmDesc()
            d.A, d.B, d.C = A.buf.ptr, B.buf.ptr, Cm.ptr
            d.sa_m, d.sa_k = 1, M
            d.sb_k, d.sb_n = N, 1
            d.ldc = d.ldc2 = N
            d.M, d.N, d.K = M, N, K
            d.splits, d.a_dtype, d.b_dtype = splits, nat.SKF_F32, nat.SKF_F32

            def run():
                rt.call('skf_gemm', nat.SKF_F64, 0, C.byref(d), ws.ptr, 512 << 20, rt.mem.stream)
            run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record(rt.mem._stream)
            for _ in range(reps):
                run()
            e1.record(rt.mem._stream)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            print('%-16s M=%d N=%d K=%d splits=%3d : %7.1f us  %5.1f TFLOP/s (dense count)' % (label, M, N, K, splits, ms * 1e3, 2.0 * M * N * K / ms / 1e9),
                  flush=True)


if __name__ == '__main__':
    main()
