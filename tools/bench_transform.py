"""Fold-in at config-3 scale: `n_new` new objects of type t1 with their relations to
the 100k objects of t2 and the 40k of t3, frozen G2 / G3 / S12 / S13 (random, as after a fit).
prepare (once per model: the two relation contractions + constant terms) and the per-iteration cost of ONE fold-in
(skf_iterate) and of the fold-ins into the models of `n_run` restarts in shared launches (skf_iterate_batch).
    python tools/bench_transform.py [n_new] [n_run]"""
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
    n_run = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    n = {'t1': n_new, 't2': 100000, 't3': 40000}
    rank = {'t1': 128, 't2': 256, 't3': 256}
    types = ['t1', 't2', 't3']
    for dtype in ('bf16', 'f32', 'f64'):
        master = 'f32' if dtype == 'bf16' else dtype
        rels = [('t1', 't2', fill_uniform((n['t1'], n['t2']), 0, dtype), None),
                ('t1', 't3', fill_uniform((n['t1'], n['t3']), 1, dtype), None)]
        plans = []
        for run in range(n_run):
            plan = DevicePlan(types, n, rank, rels, [], nat.SKF_TRANSFORM, dtype=dtype, target='t1')
            for k, t in enumerate(types):
                plan.set_factor(t, fill_uniform((n[t], rank[t]), 100 + k + 10 * run, master))
            for k, (i, j) in enumerate((('t1', 't2'), ('t1', 't3'))):
                S = fill_uniform((rank[i], rank[j]), 200 + k + 10 * run, master, scale=1e-3)
                plan.rt.call('skf_set_backbone', plan.handle, k, S.buf.ptr, S.ld, plan.rt.mem.stream)
            plans.append(plan)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plans[0].iterate(1)                          # prepare + first iteration of ONE fold-in
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        plans[0].iterate(5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plans[0].iterate(100)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 100
        t0 = time.perf_counter()
        assert DevicePlan.iterate_batch(plans, 1)    # prepare of the other models + one shared iteration
        torch.cuda.synchronize()
        prep = time.perf_counter() - t0
        DevicePlan.iterate_batch(plans, 5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        DevicePlan.iterate_batch(plans, 100)
        torch.cuda.synchronize()
        db = (time.perf_counter() - t0) / 100
        flops = sum(2.0 * n['t1'] * n[j] * rank[j] for j in ('t2', 't3'))
        print('fold-in %s: %d new objects, prepare + 1st iteration %.2f ms (%.1f TFLOP/s on the two relation '
              'contractions), then %.4f ms per iteration in ONE launch (%.0f it/s; round 1, six launches: 0.10 / 0.10 / 0.08 ms); '
              '%d models in shared launches: %.2f ms to prepare the other %d, %.4f ms per iteration of all %d'
              % (dtype, n_new, first * 1e3, flops / first / 1e12, dt * 1e3, 1.0 / dt, n_run, prep * 1e3, n_run - 1,
                 db * 1e3, n_run), flush=True)
        for plan in plans:
            plan.close()
        del rels, plans
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
