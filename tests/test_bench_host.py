"""Host-side pieces of bench.py and of the sharding tables (no GPU, no library): the algorithmic
work of BASELINE config 3, the roofline record, the self-launch of `--gpus N`, and the
partitioners' invariants."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                   # noqa: E402
import skfusion_amd                                            # noqa: E402,F401
from skfusion_amd._distributed import partition_rows, partition_relations   # noqa: E402


def test_algorithmic_work_of_config_3_and_5():
    assert bench.alg_flops(bench.FULL) == pytest.approx(9.472e12, rel=1e-12)
    n = bench.sizes(0.1)
    assert n == {'t1': 5000, 't2': 10000, 't3': 4000}
    assert bench.alg_flops(n) / bench.alg_flops(bench.FULL) == pytest.approx(0.01)
    # algorithmic work = ONE read of every relation per iteration (22 GB in bf16) and 9.472e12 flops:
    # 430 flop/B, above the 312 flop/B ridge -> the matrix-core roof binds, t_min = 3.79 ms, frac = t_min / t_kernel
    spec3 = [(i, j, False) for i, j, _ in bench.PAIRS]
    r = bench.roofline_record('bf16', bench.FULL, bench.RANKS, spec3, 12.0, 6, 9.472e12, 1, 0.015)
    assert r['alg_bytes_per_iter'] == pytest.approx(1.1e10 * 2)
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and r['peak'] == 2500.0
    assert r['t_min_ms'] == pytest.approx(9.472e12 / 2.5e15 * 1e3) and r['t_kernel_ms'] == 12.0
    assert r['frac'] == pytest.approx(r['t_min_ms'] / 12.0) == pytest.approx(r['achieved'] / r['peak'])
    assert r['achieved'] == pytest.approx(9.472e12 / 12e-3 / 1e12)                 # one iteration = 6 launches, 12 ms
    assert r['hbm_algorithmic']['achieved'] == pytest.approx(22e9 / 12e-3 / 1e9)
    # without a committed counter pass `traffic` is null and the scheduled bytes are reported beside it
    assert r['traffic'] is None and 2.0 < r['traffic_ratio'] < 2.2
    assert r['traffic_scheduled'] * 6 == pytest.approx(r['traffic_ratio'] * 22e9)
    # with the rocprofv3 --pmc passes of profiles/pmc_traffic.json (config 3 at full size only): measured bytes per launch
    pmc = bench.measured_traffic('bf16', False, 1.0)
    assert pmc is not None and 7e9 < pmc['bytes'] < 9e9 and 'FETCH_SIZE' in pmc['source']
    assert 'r03' in pmc['source']                                                    # the latest committed pass wins
    c5 = bench.measured_traffic('bf16', True, 1.0)                                   # config 5: its own pass (round 3)
    assert c5 is not None and 3e8 < c5['bytes'] < 1e9 and 'c5' in c5['source']      # (3.7e9 before the lists were pinned to XCDs)
    assert bench.measured_traffic('bf16', False, 0.1) is None and bench.measured_traffic('f32', True, 1.0) is None
    rp = bench.roofline_record('bf16', bench.FULL, bench.RANKS, spec3, 12.0, 6, 9.472e12, 1, 0.015, pmc)
    assert rp['traffic'] == pmc['bytes'] and rp['traffic_ratio'] == pytest.approx(pmc['bytes'] * 6 / 22e9)
    assert r['intensity_algorithmic'] == pytest.approx(9.472e12 / 22e9) and r['intensity_scheduled'] < r['ridge']
    assert r['mfma']['achieved'] == pytest.approx(9.472e12 / 12e-3 / 1e12) and r['mfma']['peak'] == 2500.0
    assert r['whole_iteration']['mfma_frac'] == pytest.approx(9.472e12 / 0.015 / 1e12 / 2500.0)
    r64 = bench.roofline_record('f64', bench.FULL, bench.RANKS, spec3, 180.0, 6, 9.472e12, 1, 0.19)
    assert r64['bound'] == 'mfma' and r64['unit'] == 'TFLOP/s' and r64['frac'] == pytest.approx(9.472e12 / 78.6e12 / 0.18)
    # config 5: what the launches EXECUTE (library counters), e.g. 100 launches, 3e13 flops, 4e10 relation bytes in 50 ms
    spec5 = [(i, j, d is None) for i, j, _, d in bench.C5_PAIRS]
    n5 = bench.sizes(1.0, bench.C5_FULL)
    r5 = bench.roofline_record('bf16', n5, bench.C5_RANKS, spec5, 50.0, 100, 3e13, 10, 0.1, None, 4e10, executed=True)
    assert r5['bound'] == 'mfma' and r5['t_min_ms'] == pytest.approx(12.0) and r5['frac'] == pytest.approx(0.24)
    assert r5['hbm_algorithmic']['achieved'] == pytest.approx(800.0) and r5['traffic'] is None
    assert bench.physical_cores() is None or bench.physical_cores() >= 1
    spec = [(i, j, d is None) for i, j, _, d in bench.C5_PAIRS]
    n5 = bench.sizes(1.0, bench.C5_FULL)
    base = sum(2.0 * n5[i] * n5[j] * (bench.C5_RANKS[i] + bench.C5_RANKS[j]) for i, j, _ in spec)
    extra = 2.0 * n5['user'] * n5['movie'] * min(bench.C5_RANKS['user'], bench.C5_RANKS['movie'])
    assert bench.alg_flops(n5, spec, bench.C5_RANKS) == pytest.approx(base + extra)


@pytest.mark.parametrize('size', [1, 2, 3, 4, 5, 8, 16])
@pytest.mark.parametrize('graph', ['c3', 'c5', 'tiny'])
def test_row_partition_covers_every_row_once_and_balances(size, graph):
    if graph == 'c3':
        n, c, align = bench.FULL, bench.RANKS, 256
        rel = [(i, j, None, None) for i, j, _ in bench.PAIRS]
    elif graph == 'c5':
        n, c, align = bench.C5_FULL, bench.C5_RANKS, 256
        rel = [(i, j, None, None) for i, j, _, _ in bench.C5_PAIRS]
    else:
        n, c, align = {'a': 12, 'b': 9, 'c': 7}, {'a': 4, 'b': 3, 'c': 2}, 1
        rel = [('a', 'b', None, None), ('a', 'c', None, None), ('b', 'c', None, None)]
    thetas = [(rel[0][0], None)]
    blocks, th_owner = partition_rows(rel, thetas, n, c, align=align, size=size)
    load = np.zeros(size)
    for (i, j, _, _), blk in zip(rel, blocks):
        pos, seen = 0, set()
        for q, a, cnt in blk:
            assert a == pos and cnt > 0 and 0 <= q < size      # contiguous, in row order
            assert q not in seen                                 # at most one block per rank and relation
            seen.add(q)
            assert a % align == 0
            pos += cnt
            load[q] += cnt * n[j] * (c[i] + c[j])
        assert pos == n[i]                                       # every row exactly once
        assert blk[0][1] == 0                                    # the first block owns the column side
    assert 0 <= th_owner[0] < size
    if graph != 'tiny' and size <= 8:
        assert load.max() <= 1.08 * load.sum() / size           # config 3 on 8 GPUs: 1.033


def test_relation_partition_is_deterministic_and_complete():
    rel = [(i, j, None, None) for i, j, _ in bench.PAIRS]
    import skfusion_amd._distributed as D
    saved = D.world
    try:
        D.world = lambda: (0, 2)
        a = partition_relations(rel, [('t1', None)], bench.FULL, bench.RANKS)
        b = partition_relations(rel, [('t1', None)], bench.FULL, bench.RANKS)
    finally:
        D.world = saved
    assert a == b and sorted(set(a[0])) == [0, 1] and len(a[0]) == 3


def test_bench_gpus_n_launches_itself_and_checks_the_world_size(monkeypatch):
    """`python bench.py --gpus 2` without a launcher re-executes itself under torch.distributed.run with one
    rank per GPU; under a launcher WORLD_SIZE must equal --gpus (round-1 finding: --gpus was never read)."""
    import subprocess
    calls = []
    monkeypatch.setattr(subprocess, 'call', lambda cmd: calls.append(cmd) or 0)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '2', '--steps', '3', '--scale', '0.2'])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and len(calls) == 1
    cmd = calls[0]
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1']
    assert cmd[cmd.index('--nproc-per-node') + 1] == '2' and cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[-6:] == ['--gpus', '2', '--steps', '3', '--scale', '0.2'] and cmd[-7].endswith('bench.py')
    # under a launcher: a mismatch is refused before any GPU work
    monkeypatch.setenv('WORLD_SIZE', '4')
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert 'WORLD_SIZE=4' in str(e.value.code)


def test_parity_full_size_record_plumbing(tmp_path):
    """`parity_full_size` of the bench record (VERDICT rounds 3 / 4, north_star: same inputs, results matching the reference
    path): the oracle leg keeps its backbones and 64 rows of every factor after iterations 2 AND 5 (PARITY_CHECKPOINTS) and
    the relation errors after the last (`_oracle_timing(keep=...)`, what `--cpu-full-child --parity-out` writes at full
    size); `parity_record` compares an engine's side with them and gates the backbones by the conditioning of the Gram
    matrices they were formed from.  Here at 1/100 scale with the oracle's own two-GEMM form standing in for the engine: every
    key present, deviations at rounding level, the gate open; a perturbed engine side shows up in the matching key, and a
    backbone off by more than cond_i cond_j eps closes the gate."""
    from oracle import dfmf_oracle as orc
    assert bench.PARITY_ITERS >= 5 and bench.PARITY_CHECKPOINTS[-1] == bench.PARITY_ITERS
    path = str(tmp_path / 'parity.npz')
    times, n = bench._oracle_timing(0.01, bench.PARITY_ITERS, keep=path)
    assert len(times) == bench.PARITY_ITERS and n == {'t1': 500, 't2': 1000, 't3': 400}
    R = {(i, j): [orc.hash_uniform_matrix(s, n[i], n[j])] for i, j, s in bench.PAIRS}
    G = {(t, t): orc.hash_uniform_matrix(100 + k, n[t], bench.RANKS[t]) for k, t in enumerate(bench.TYPES)}
    eng = {}
    for it in range(1, bench.PARITY_ITERS + 1):
        if it in bench.PARITY_CHECKPOINTS:
            for t in bench.TYPES:
                eng['kappa_%s@%d' % (t, it)] = bench.gram_condition(G[t, t])
        G, S = orc.dfmf_two_gemm_step(R, G, {}, {})
        if it in bench.PARITY_CHECKPOINTS:
            for i, j, _ in bench.PAIRS:
                eng['S_%s_%s@%d' % (i, j, it)] = S[i, j][0]
            for t in bench.TYPES:
                rows = bench.parity_rows(n[t])
                assert len(rows) == bench.PARITY_ROWS and rows[0] == 0 and rows[-1] == n[t] - 1
                eng['G_%s@%d' % (t, it)] = G[t, t][rows]
    errs = orc.relation_errors(R, G, S)
    for i, j, _ in bench.PAIRS:
        eng['err_%s_%s' % (i, j)] = errs[i, j][0]
    assert eng['kappa_t1@2'] > 100.0               # uniform random factors: cond(G^T G) ~ 1 + 3 c
    rec = bench.parity_record(path, eng, 'f64')
    assert set(rec) >= {'iters', 'checkpoints', 'S_relerr', 'G_rows_relerr', 'err_relerr', 'oracle_err', 'S_gate'}
    assert rec['iters'] == bench.PARITY_ITERS and sorted(rec['checkpoints']) == sorted(str(c) for c in bench.PARITY_CHECKPOINTS)
    assert rec['S_relerr'] < 1e-8 and rec['G_rows_relerr'] < 1e-10 and rec['err_relerr'] < 1e-12
    assert rec['S_gate']['ok'] and rec['S_gate']['eps'] == bench.PARITY_S_EPS['f64']
    last = str(bench.PARITY_ITERS)
    assert rec['checkpoints'][last]['S_relerr_over_conditioning'] <= rec['checkpoints'][last]['S_relerr'] / 1e4
    eng['G_t2@%s' % last] = eng['G_t2@%s' % last] * (1.0 + 1e-3)
    eng['err_t1_t3'] = eng['err_t1_t3'] * (1.0 - 2e-4)
    bad = bench.parity_record(path, eng, 'f64')
    assert bad['G_rows_relerr'] == pytest.approx(1e-3, rel=1e-6) and bad['err_relerr'] == pytest.approx(2e-4, rel=1e-6)
    assert bad['S_relerr'] == rec['S_relerr'] and bad['S_gate']['ok']
    # a backbone that is off by more than its conditioning explains: the gate closes (f64 eps), and opens again at bf16's
    eng['S_t1_t2@2'] = eng['S_t1_t2@2'] * (1.0 + 1e-4)
    assert not bench.parity_record(path, eng, 'f64')['S_gate']['ok']
    assert bench.parity_record(path, eng, 'bf16')['S_gate']['ok'] == (1e-4 / (eng['kappa_t1@2'] * eng['kappa_t2@2']) <= bench.PARITY_S_EPS['bf16'])


def _fake_run(calls, fail=()):
    def run(workload, dtype, steps, warmup, scale=1.0, data='uniform', mode='restarts', rank=0, world=1, dist=None,
            backend='nccl', emulate=None, parity=False, sustained=0):
        calls.append(dict(workload=workload, mode=mode, rank=rank, world=world, emulate=emulate, steps=steps))
        if workload in fail:
            raise RuntimeError('no such luck on %s' % workload)
        return {'elapsed': 0.002 * steps, 'k_ms': 1.5 * steps, 'exchange_bytes': 178.7e6 if workload == 'c3' else 190e6,
                'launches_per_step': 61.0, 'enqueue_ms_per_step': 0.4, 'rmse': {'t1-t2': 0.2887},
                'comm': None if emulate else {'rank': rank, 'world': world, 'transport': 'rccl', 'transport_ranks': world}}
    return run


def test_rank_of_8_sub_record_of_the_default_line(monkeypatch):
    """the default one-GPU line carries `workloads.rank_of_8` -- rank 3 of 8 of the ownership-sharded
    fit on configs 3 and 5 with the exchanges skipped: ms, launches, exchange bytes, modelled wire time; a leg that fails
    reports its error and leaves the other standing."""
    calls = []
    monkeypatch.setattr(bench, 'run_workload', _fake_run(calls))
    rec = bench.rank_of_8_record('bf16')
    assert [c['workload'] for c in calls] == ['c3', 'c5'] and all(c['emulate'] == (3, 8) for c in calls)
    for key, nbytes in (('c3', 178.7e6), ('c5', 190e6)):
        leg = rec[key]
        assert leg['compute_ms_per_step'] == pytest.approx(2.0) and leg['launches_per_step'] == 61.0
        assert leg['exchange_bytes_per_rank_and_iter'] == nbytes and leg['contraction_ms_per_step'] == pytest.approx(1.5)
        assert leg['wire_ms_ring'] == pytest.approx(nbytes / 153e9 * 1e3) and leg['wire_ms_all_links'] == pytest.approx(leg['wire_ms_ring'] / 7)
    assert rec['rank'] == 3 and rec['world'] == 8 and 'exchanges skipped' in rec['what']
    calls[:] = []
    monkeypatch.setattr(bench, 'run_workload', _fake_run(calls, fail=('c5',)))
    rec = bench.rank_of_8_record('bf16')
    assert 'compute_ms_per_step' in rec['c3'] and 'no such luck' in rec['c5']['error']


def test_strong_sub_record_of_an_n_gpu_run(monkeypatch):
    """an N > 1 run in the default mode (restarts) also runs ONE fit sharded by ownership over the same
    process group on configs 3 and 5 and reports it as `strong`: it/s, bytes per rank and iteration, the transport and the
    ranks it reports (RCCL: ncclCommCount)."""
    calls = []
    monkeypatch.setattr(bench, 'run_workload', _fake_run(calls))
    rec = bench.strong_record('bf16', 0, 4, dist=object(), backend='nccl')
    assert [(c['workload'], c['mode'], c['world']) for c in calls] == [('c3', 'owned', 4), ('c5', 'owned', 4)]
    assert rec['scaling'] == 'strong' and rec['mode'] == 'owned' and rec['n_gpus'] == 4 and rec['backend'] == 'nccl'
    for key in ('c3', 'c5'):
        leg = rec[key]
        assert leg['value'] == pytest.approx(500.0) and leg['unit'] == 'iters/s' and leg['ms_per_step'] == pytest.approx(2.0)
        assert leg['transport'] == 'rccl' and leg['transport_ranks'] == 4 and leg['exchange_bytes_per_rank_and_iter'] > 1e8


def test_bounded_leg_returns_or_times_out():
    """The `strong` leg of an N > 1 run is bounded in time (a collective that never returns must not cost the restarts line)."""
    import time
    res, hung = bench.run_bounded(lambda: {'ok': 1}, 5.0)
    assert res == {'ok': 1} and not hung
    res, hung = bench.run_bounded(lambda: time.sleep(5.0), 0.2)
    assert hung and 'timed out' in res['error']
    res, hung = bench.run_bounded(lambda: 1 / 0, 5.0)
    assert not hung and 'division' in res['error']


@pytest.mark.parametrize('ending', ['dies', 'returns', 'terminated'])
def test_line_watchdog_prints_the_line_only_when_rank_0_does_not_come_back(ending, tmp_path):
    """Round 6: the measured line of an N > 1 run survives a rank 0 that dies inside the `strong` leg
    (a fault inside a collective takes the whole process) -- the forked watchdog prints it; a rank 0 that comes back
    disarms the watchdog and prints the full line itself: ONE line either way."""
    import json
    import subprocess
    body = {
        'dies': "os.kill(os.getpid(), signal.SIGKILL)",
        'returns': "w.disarm(); print(json.dumps({'value': 1, 'strong': {'ok': 1}}), flush=True)",
        'terminated': "w.disarm(); print(json.dumps({'value': 1, 'interrupted': 'signal'}), flush=True); os._exit(0)",
    }[ending]
    code = ("import os, sys, json, signal\n"
            "sys.path.insert(0, %r)\n"
            "import bench\n"
            "w = bench.LineWatchdog(json.dumps({'value': 1, 'strong': {'error': 'rank 0 died inside the strong leg'}}))\n"
            "%s\n" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), body))
    out = tmp_path / 'out.txt'
    with open(out, 'wb') as fh:
        proc = subprocess.run([sys.executable, '-c', code], stdout=fh, stderr=subprocess.PIPE, timeout=120)
    import time
    for _ in range(50):                       # the watchdog of a killed parent prints a moment after the parent is gone
        lines = [ln for ln in out.read_text().splitlines() if ln.startswith('{')]
        if lines:
            break
        time.sleep(0.1)
    assert len(lines) == 1, (lines, proc.stderr[-300:])
    rec = json.loads(lines[0])
    assert rec['value'] == 1
    if ending == 'dies':
        assert proc.returncode == -9 and 'died' in rec['strong']['error']
    elif ending == 'returns':
        assert rec['strong'] == {'ok': 1}
    else:
        assert 'interrupted' in rec
