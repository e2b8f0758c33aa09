"""Diagnostics of the planted workload's device generator (bench.c3_relation, data='planted') at growing sizes: statistics
of the relation, rows against the host formula, and a short fit.   python tools/diag_planted.py [scales...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    import skfusion_amd._native as nat
    from skfusion_amd._engine import DevicePlan, fill_uniform
    from oracle import dfmf_oracle as orc
    scales = [float(v) for v in sys.argv[1:]] or [0.04, 0.3, 1.0]
    floor = 0.01 / np.sqrt(12.0)
    for scale in scales:
        n = bench.sizes(scale)
        for dtype in ('f32', 'bf16'):
            cache = {}
            rels = []
            for k, (i, j, seed) in enumerate(bench.PAIRS):
                dm = bench.c3_relation(k, n, dtype, 'planted', cache)
                t = dm.buf.owner
                rows = np.array([0, 1, n[i] // 2, n[i] - 1])
                got = t[torch.from_numpy(rows).cuda()].to(torch.float64).cpu().numpy()
                # host formula on those rows (mean taken from the device tensor: the host cannot form the whole matrix)
                Gi = orc.hash_uniform_at(200 + bench.TYPES.index(i), rows[:, None] * bench.RANKS[i] + np.arange(bench.RANKS[i])[None, :])
                Gj = orc.hash_uniform_matrix(200 + bench.TYPES.index(j), n[j], bench.RANKS[j])
                S = orc.hash_uniform_matrix(300 + seed, bench.RANKS[i], bench.RANKS[j])
                raw = Gi @ S @ Gj.T
                noise = orc.hash_uniform_at(400 + seed, rows[:, None] * n[j] + np.arange(n[j])[None, :])
                # mean of the raw product = mean(G*_i)^T S mean(G*_j) summed: E[raw] computed exactly from column means
                mi = orc.hash_uniform_matrix(200 + bench.TYPES.index(i), n[i], bench.RANKS[i]).mean(axis=0)
                mean = float(mi @ S @ Gj.mean(axis=0))
                want = raw / mean + 0.01 * noise
                tf = t.to(torch.float32) if dtype == 'bf16' else t
                stats = (float(tf[:4096].mean()), float(tf[:4096].std()), float(tf[:4096].min()), float(tf[:4096].max()))
                print('scale %.2f %s rel %d (%dx%d): rows vs host formula relerr %.3e; first 4096 rows mean %.4f std %.4f min %.4f max %.4f; quant %s'
                      % (scale, dtype, k, n[i], n[j], np.linalg.norm(got - want) / np.linalg.norm(want), stats[0], stats[1], stats[2],
                         stats[3], cache.get('quant_%d' % k)), flush=True)
                rels.append((i, j, dm, None))
            cache.clear()
            plan = DevicePlan(bench.TYPES, n, bench.RANKS, rels, [], nat.SKF_DFMF, dtype=dtype)
            plan.release_relation_data()
            del rels, dm, t, tf
            torch.cuda.empty_cache()
            for k, t_ in enumerate(bench.TYPES):
                plan.set_factor(t_, fill_uniform((n[t_], bench.RANKS[t_]), 100 + k, 'f32'))
            for its in (5, 25, 30):
                plan.iterate(its)
                r = [np.sqrt(plan.relation_sqerr(k) / (float(n[i]) * n[j])) / floor for k, (i, j, _) in enumerate(bench.PAIRS)]
                print('   after +%d iterations: RMSE / floor %s' % (its, ' '.join('%.4f' % v for v in r)), flush=True)
            plan.close()
            torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
