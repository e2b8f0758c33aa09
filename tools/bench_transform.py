"""Fold-in at config-3 scale: `n_new` new objects of type t1 with their relations to
the 100k objects of t2 and the 40k of t3, frozen G2 / G3 / S12 / S13 (random, as after a fit), f32
engine.  prepare (once: the two relation contractions + constant terms) and the per-iteration cost.
    python tools/bench_transform.py [n_new]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import skfusion_amd._native as nat
    from skfusion_amd._engine import DevicePlan, fill_uniform
    n_new = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    n = {'t1': n_new, 't2': 100000, 't3': 40000}
    rank = {'t1': 128, 't2': 256, 't3': 256}
    types = ['t1', 't2', 't3']
    for dtype in ('bf16', 'f32', 'f64'):
        master = 'f32' if dtype == 'bf16' else dtype
        rels = [('t1', 't2', fill_uniform((n['t1'], n['t2']), 0, dtype), None),
                ('t1', 't3', fill_uniform((n['t1'], n['t3']), 1, dtype), None)]
        plan = DevicePlan(types, n, rank, rels, [], nat.SKF_TRANSFORM, dtype=dtype, target='t1')
        for k, t in enumerate(types):
            plan.set_factor(t, fill_uniform((n[t], rank[t]), 100 + k, master))
        for k, (i, j) in enumerate((('t1', 't2'), ('t1', 't3'))):
            S = fill_uniform((rank[i], rank[j]), 200 + k, master, scale=1e-3)
            plan.rt.call('skf_set_backbone', plan.handle, k, S.buf.ptr, S.ld, plan.rt.mem.stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan.iterate(1)                              # prepare + first iteration
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        t0 = time.perf_counter()
        plan.iterate(50)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50
        flops = sum(2.0 * n['t1'] * n[j] * rank[j] for j in ('t2', 't3'))
        print('fold-in %s: %d new objects, prepare + 1st iteration %.2f ms (%.1f TFLOP/s on the two relation '
              'contractions), then %.3f ms per iteration (%.0f it/s)'
              % (dtype, n_new, first * 1e3, flops / first / 1e12, dt * 1e3, 1.0 / dt), flush=True)
        plan.close()
        del rels
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
