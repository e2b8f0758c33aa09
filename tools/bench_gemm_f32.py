"""Stand-alone timing of the strided MFMA GEMM (skf_gemm) in f32 / f64."""
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
    cases = [('sq4096 NN', 4096, 4096, 4096, False), ('sq4096 TN', 4096, 4096, 4096, True),
             ('P12 NN', 50000, 256, 100000, False), ('Q12 TN', 100000, 128, 50000, True),
             ('side NN', 100000, 256, 256, False)]
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtypes', default='f32,f64')
    ap.add_argument('--cases', default='', help='comma-separated first words of the case labels (sq4096, P12, Q12, side); default all')
    ap.add_argument('--reps', type=int, default=3)
    args = ap.parse_args()
    want = [c for c in args.cases.split(',') if c]
    for dt, name in ((nat.SKF_F32, 'f32'), (nat.SKF_F64, 'f64')):
        if name not in args.dtypes.split(','):
            continue
        for label, M, N, K, transA in cases:
            if want and label.split()[0] not in want:
                continue
            if dt == nat.SKF_F64 and M * K > 3e9:
                continue
            es = 4 if dt == nat.SKF_F32 else 8
            A = fill_uniform((K, M) if transA else (M, K), 1, name)
            B = fill_uniform((K, N), 2, name)
            Cm = rt.mem.empty(M * N * es)
            d = nat.GemmDesc()
            d.A, d.B, d.C = A.buf.ptr, B.buf.ptr, Cm.ptr
            d.sa_m, d.sa_k = (1, M) if transA else (K, 1)
            d.sb_k, d.sb_n = N, 1
            d.ldc = d.ldc2 = N
            d.M, d.N, d.K = M, N, K
            d.splits, d.a_dtype, d.b_dtype = 1, -1, -1

            def run():
                rt.call('skf_gemm', dt, 0, C.byref(d), None, 0, rt.mem.stream)
            run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = args.reps
            e0.record(rt.mem._stream)
            for _ in range(reps):
                run()
            e1.record(rt.mem._stream)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            print('%s %-10s M=%d N=%d K=%d : %.3f ms  %.1f TFLOP/s' % (name, label, M, N, K, ms, 2.0 * M * N * K / ms / 1e9),
                  flush=True)
            del A, B, Cm


if __name__ == '__main__':
    main()
