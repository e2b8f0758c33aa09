"""Stand-alone timing of the bf16 relation contraction (skf_gemm_bf16) on the config-3 shapes.
    python tools/bench_gemm_bf16.py [--reps 10] [--shapes P12,Q12,...] [--tiles 128,256] [--splits 0,9]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = {'P12': (50000, 256, 100000), 'Q12': (100000, 128, 50000), 'P13': (50000, 256, 40000),
          'Q13': (40000, 128, 50000), 'P23': (100000, 256, 40000), 'Q23': (40000, 256, 100000)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--shapes', default='P12,Q12')
    ap.add_argument('--tiles', default='128,256')
    ap.add_argument('--splits', default='0')
    ap.add_argument('--tn', action='store_true', help='transposed-A form (skf_gemm_bf16_tn): A = the row-major relation [K][M]')
    ap.add_argument('--zero', action='store_true', help='all-zero operands (power / DVFS probe)')
    args = ap.parse_args()
    import torch
    import skfusion_amd._native as nat
    from skfusion_amd._engine import fill_uniform
    rt = nat.get_runtime()
    for name in args.shapes.split(','):
        M, N, K = SHAPES[name]
        Kp = (K + 63) // 64 * 64
        mp, npad = (M + 255) // 256 * 256, (N + 255) // 256 * 256
        A = fill_uniform((mp, Kp), 1, 'bf16')
        Bt = fill_uniform((npad, Kp), 2, 'bf16')
        if args.zero:
            A = fill_uniform((mp, Kp), 1, 'bf16', scale=0.0)
            Bt = fill_uniform((npad, Kp), 2, 'bf16', scale=0.0)
        C = rt.mem.empty(M * N * 4)
        ws = rt.mem.empty(32 * M * N * 4)
        if args.tn:
            lda = (M + 63) // 64 * 64
            A = fill_uniform((Kp, lda), 1, 'bf16', scale=0.0 if args.zero else 1.0)
        for tile in args.tiles.split(','):
            for sp in args.splits.split(','):
                sp = int(sp)

                def run():
                    if args.tn:
                        rt.call('skf_gemm_bf16_tn', A.buf.ptr, lda, Bt.buf.ptr, Kp, C.ptr, N, M, N, Kp, sp, ws.ptr,
                                ws.nbytes, rt.mem.stream)
                    else:
                        rt.call('skf_gemm_bf16', A.buf.ptr, Kp, Bt.buf.ptr, Kp, C.ptr, N, M, N, Kp, sp, ws.ptr,
                                ws.nbytes, rt.mem.stream)
                run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(rt.mem._stream)
                for _ in range(args.reps):
                    run()
                e1.record(rt.mem._stream)
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / args.reps
                print(('zero ' if args.zero else '') + ('tn ' if args.tn else '') + '%s M=%d N=%d K=%d tile=%s splits=%d : %.3f ms  %.0f TFLOP/s  (A stream %.2f TB/s)'
                      % (name, M, N, K, tile, sp, ms, 2.0 * M * N * K / ms / 1e9, M * Kp * 2 / ms / 1e9), flush=True)


if __name__ == '__main__':
    main()
