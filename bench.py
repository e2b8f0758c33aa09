#!/usr/bin/env python
"""bench.py -- DFMF update iterations/sec + reconstruction RMSE on the synthetic 3-type graph of
BASELINE.json (configs[2]: 50k x 100k / 50k x 40k / 100k x 40k, ranks 128/256/256).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full DFMF iteration (Gram + pinv, S closed form, the two relation contractions
per relation, the multiplicative G update) over the whole graph, data resident in HBM
(generated on the device by the counter-based generator shared with the oracle).  With N > 1
every rank runs an independent random restart of the same graph (BASELINE config 4, the
joblib n_run loop of reference dfmf.py:87-95): no data-path collective, weak scaling,
value = N * K / max-over-ranks time.

With --gpus N > 1 and no WORLD_SIZE in the environment the script launches itself under
torch.distributed.run (one rank per GPU); under a launcher it checks WORLD_SIZE == --gpus.

The JSON line also carries
  roofline     : the dominant kernel = the relation contractions P = R G_j, Q = R^T G_i, timed with hipEvents on the
                 launch stream inside the timed region.  SURVEY.md 8(d): the algorithmic work of an iteration is ONE read of
                 every relation (22 GB in bf16) and sum 2 n_i n_j (c_i + c_j) flops = 9.472e12 -- 430 flop/B, above the
                 312 flop/B ridge, so the MATRIX-CORE roof binds: t_min = max(flops / peak, bytes / 8 TB/s) = 3.79 ms per
                 iteration, and frac = t_min / t_kernel (= achieved / peak of the binding roof).  Beside it: `hbm_algorithmic`
                 (one-read bytes / launch time), `hbm_scheduled` (the engine reads every relation once per contraction,
                 i.e. twice per iteration: a fused P+Q pass would have to spill one output as partial sums), `traffic` =
                 HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE children of THIS run (`traffic_kind`:
                 "PMC, this run ..."; the committed pass of profiles/pmc_traffic.json stands in where the box has no
                 rocprofv3, labelled so; `pmc_errors` says why a pass did not deliver), `traffic_scheduled`, `whole_iteration`.
  sustained    : a second timed region of 1000 steps on the same plan.
  parity_full_size : per engine, its first FIVE iterations against the oracle's at FULL size on the same inputs (checkpoints
                 after iterations 2 and 5: backbones, 64 rows of every factor; relation errors after 5), and `S_gate`: the
                 backbone deviation over cond(Gram_i) cond(Gram_j) of the engine's own factors against eps(engine).
  engines      : short runs (3 steps) of the f32 and f64 engines on the same graph (the reference computes in f64).
  workloads    : the other single-GPU BASELINE configurations, default run only --
                   c5_dfmc    configs[4]: Dfmc on the MovieLens-style graph (10 steps), its own roofline (flops the launches
                              EXECUTE and relation bytes as STORED: bitmaps, gathers, known-entry lists; traffic from PMC
                              children of this run) and cpu_baseline (oracle dfmc MEASURED at 1/2 linear scale: `scale: 0.5`,
                              `projection: false`; the 1/4-scale projection only where the child cannot run);
                   c2_dicty   configs[1]: the dicty graph (tests/golden/dicty_inputs.npz), 100 iterations f32 and f64,
                              and the NumPy oracle on the same host;
                   c3_planted configs[2] on planted data (rank-structured + 1 % noise): RMSE after 30 iterations over the
                              noise floor 0.01 / sqrt(12);
                   rank_of_8  rank 3 of 8 of the fit sharded by ownership, exchanges skipped, configs[2] and configs[4]:
                              compute ms, launches, exchange bytes per iteration and their modelled wire time;
                   c3_tenth   configs[2] at 1/10 linear scale in the three engines, the oracle's rate beside them.
  strong       : N > 1 runs in the default mode only -- after the restarts measurement the SAME process group runs ONE fit
                 sharded by ownership (--mode owned) on configs[2] and configs[4]: it/s, bytes per rank and iteration, the
                 transport and the ranks it reports (RCCL: ncclCommCount).  Bounded by SKF_STRONG_TIMEOUT (300 s): a
                 collective that never returns costs this sub-record, not the line.
  cpu_baseline : the NumPy oracle (reference operation order, fp64, all BLAS threads): FIVE iterations at FULL size when the
                 host can hold the 88 GB of fp64 relations (child process, bounded; it runs beside the GPU-only sub-records
                 of the default run), the last one reported; inputs from
                 the counter-based generator the device uses; otherwise a 1/10-linear-scale sample scaled by the work ratio
                 (`projection: true`).

Options beyond the driver's contract (defaults = the metric's configuration):
  --dtype bf16|f32|f64      engine (default bf16, the configuration the metric is quoted on)
  --mode restarts|relations|rows|owned   N > 1: independent restarts (default, weak scaling) or ONE fit
                            sharded by whole relations / balanced row blocks (strong scaling,
                            RCCL all-reduces between the stages) / ownership of the factor rows (strong
                            scaling with the least exchange: reduce-scatter of each partial Q, all-gather
                            of the updated rows -- DESIGN.md 7)
  --emulate-rank k/W|all/W  ONE GPU: the compute of rank k (or of every rank in turn) of a fit sharded by
                            ownership over W ranks, exchanges skipped (null communicator) -- the measured
                            half of the multi-GPU critical path of DESIGN.md 7; adds `emulated_ranks`
  --workload c3|c5          c5 = BASELINE configs[4]: Dfmc, MovieLens-style 6-relation graph
  --data uniform|planted    planted: rank-structured relations + 1 % noise (RMSE discriminates)
  --scale x                 linear scale of the object counts (smoke runs)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FULL = {'t1': 50000, 't2': 100000, 't3': 40000}
RANKS = {'t1': 128, 't2': 256, 't3': 256}
TYPES = ['t1', 't2', 't3']
PAIRS = [('t1', 't2', 0), ('t1', 't3', 1), ('t2', 't3', 2)]     # (row, col, data seed)
PEAK_TFLOPS = {'f32': 157.3, 'f64': 78.6, 'bf16': 2500.0}        # MI355X dense matrix peaks
MASTER = {'f32': 'f32', 'f64': 'f64', 'bf16': 'f32'}


# BASELINE configs[4]: MovieLens-style Dfmc -- 6 object types, 6 relations, the ratings
# relation 98 % masked, constraints on Movie (lambda*I and a sparse negative similarity).  The dense
# lambda*I on the 100k users (40 GB as a dense f32 matrix, as the reference would hold it) is left out.
C5_FULL = {'user': 100000, 'movie': 40000, 'genre': 64, 'actor': 40000, 'tag': 20000, 'director': 10000}
C5_RANKS = {'user': 128, 'movie': 256, 'genre': 16, 'actor': 128, 'tag': 64, 'director': 64}
C5_TYPES = ['user', 'movie', 'genre', 'actor', 'tag', 'director']
# (row, col, seed, density of the binary relation | None = ratings, masked)
C5_PAIRS = [('user', 'movie', 50, None), ('movie', 'genre', 51, 0.15), ('movie', 'actor', 52, 0.001),
            ('movie', 'tag', 53, 0.05), ('movie', 'director', 54, 0.001), ('user', 'tag', 55, 0.01)]


def sizes(scale, full=None):
    return {t: max(int(round(n * scale)), 64) for t, n in (full or FULL).items()}


def c5_graph(n, dtype, masked=0.98, lam=0.01):
    """Device-resident synthetic data of the config-5 graph (torch used as the random source and
    elementwise plumbing; the engine only sees raw device pointers)."""
    import torch
    from skfusion_amd._engine import device_matrix_from_tensor as wrap
    gen = torch.Generator(device='cuda')
    tdt = {'bf16': torch.bfloat16, 'f32': torch.float32, 'f64': torch.float64}[dtype]
    mdt = {'bf16': torch.float32, 'f32': torch.float32, 'f64': torch.float64}[dtype]
    rels = []
    for i, j, seed, dens in C5_PAIRS:
        gen.manual_seed(seed)
        u = torch.rand((n[i], n[j]), generator=gen, device='cuda', dtype=torch.float32)
        if dens is None:
            data = ((u * 10.0).floor_() + 1.0).div_(10.0).to(tdt)             # ratings 0.1 .. 1.0
            gen.manual_seed(seed + 100)
            mask = (torch.rand((n[i], n[j]), generator=gen, device='cuda', dtype=torch.float32) < masked).to(torch.uint8)
            mm = wrap(mask.contiguous())
            mm.known = int(mask.numel() - int(mask.sum(dtype=torch.int64).item()))   # few known entries: kept as lists (known_bound)
            rels.append((i, j, wrap(data.contiguous()), mm))
        else:
            dm = wrap((u < dens).to(tdt).contiguous())
            dm.binary = True                       # 0 / 1 relation: the bf16 engine keeps it as a bitmap (SKF_REL_BINARY)
            rels.append((i, j, dm, None))
        del u
    nm = n['movie']
    gen.manual_seed(60)
    # ~2 similar movies per movie whatever the scale, weight -0.001: lambda*I + sim stays diagonally
    # dominant (positive semi-definite).  A must-link constraint that outweighs lambda rewards large G
    # (tr(G^T Theta G) < 0) where 98 % masked ratings leave the movie factor unconstrained: measured at
    # full size, the factor's norm then grows x20 per iteration (density 1e-3, weight -0.05) until the
    # Gram matrix is numerically singular and the pseudo-inverse falls back to the slow eigen path
    sim = torch.rand((nm, nm), generator=gen, device='cuda', dtype=torch.float32) < 1.0 / nm
    sim = (sim | sim.t()).to(mdt).mul_(-0.001)
    sim.fill_diagonal_(0.0)
    eye = torch.zeros((nm, nm), device='cuda', dtype=mdt)
    eye.fill_diagonal_(lam)
    thetas = [('movie', wrap(eye)), ('movie', wrap(sim.contiguous()))]
    thetas[0][1].nnz = nm                                     # sparse constraints: the engine keeps them as CSR
    thetas[1][1].nnz = int((sim != 0).sum().item())
    torch.cuda.synchronize()
    return rels, thetas


def c3_relation(k, n, dtype, data='uniform', cache=None):
    """Relation k of the config-3 graph in HBM.  uniform: iid U[0,1) from the counter-based generator shared with
    the oracle (regenerable on the host, element by element).  planted: R_ij = G*_i S*_ij G*_j^T / mean + 0.01 U
     with G*, S*, U from the same counter-based generator (seeds 200 + type, 300 + relation, 400 + relation:
    the graph of tests/helpers.py:c3_planted_graph and of the reference-derived golden c3_planted_scaled.npz, formed here in
    f32 on the device; torch as data plumbing only) -- noise floor 0.01 / sqrt(12) = 0.0029.  `cache` (a dict) keeps G*
    between the relations and receives 'quant_k' = ||bf16(R) - R||_F / sqrt(cells) of a bf16 relation."""
    from skfusion_amd._engine import fill_uniform
    i, j, seed = PAIRS[k]
    if data == 'uniform':
        return fill_uniform((n[i], n[j]), seed, dtype)
    import torch
    from skfusion_amd._engine import device_matrix_from_tensor as wrap
    cache = cache if cache is not None else {}

    def hashed(shape, sd):
        # (the engine fills on ITS stream: torch's pending work first -- the caching allocator may hand out a block that a
        # queued torch kernel still reads -- and the fill before torch touches the result)
        torch.cuda.synchronize()
        dm = fill_uniform(shape, sd, 'f32')
        torch.cuda.synchronize()
        return dm.buf.owner[:shape[0] * shape[1] * 4].view(torch.float32).view(shape)
    for q, t in enumerate(TYPES):
        if t not in cache:
            cache[t] = hashed((n[t], RANKS[t]), 200 + q)
    Rm = (cache[i] @ hashed((RANKS[i], RANKS[j]), 300 + seed)) @ cache[j].t()
    Rm.div_(Rm.mean(dtype=torch.float64).to(torch.float32))
    noise = hashed((n[i], n[j]), 400 + seed)
    for r0 in range(0, n[i], 8192):          # (row chunks: no third full-size temporary)
        Rm[r0:r0 + 8192].add_(noise[r0:r0 + 8192], alpha=0.01)
    del noise
    tdt = {'bf16': torch.bfloat16, 'f32': torch.float32, 'f64': torch.float64}[dtype]
    out_t = Rm.to(tdt).contiguous()
    if dtype == 'bf16':
        sq = 0.0
        for r0 in range(0, n[i], 8192):
            d = (out_t[r0:r0 + 8192].to(torch.float32) - Rm[r0:r0 + 8192]).to(torch.float64)
            sq += float((d * d).sum().item())
        cache['quant_%d' % k] = float(np.sqrt(sq / (float(n[i]) * n[j])))
    out = wrap(out_t)
    torch.cuda.synchronize()
    return out


def alg_flops(n, spec=None, ranks=None):
    """sum over relations of 2 * n_i * n_j * (c_i + c_j); a masked relation (DFMC) adds
    its completion 2 * n_i * n_j * min(c_i, c_j)."""
    spec = spec or [(i, j, False) for i, j, _ in PAIRS]
    ranks = ranks or RANKS
    return sum(2.0 * n[i] * n[j] * (ranks[i] + ranks[j] + (min(ranks[i], ranks[j]) if m else 0))
               for i, j, m in spec)


HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable with a float4 copy)
RIDGE = {'bf16': 2500e12 / 8e12, 'f32': 157.3e12 / 8e12, 'f64': 78.6e12 / 8e12}     # flop per HBM byte


def measured_traffic(dtype, c5, scale):
    """HBM bytes per contraction launch from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json: FETCH_SIZE and
    WRITE_SIZE in passes of their own over this very command, FETCH_SIZE with the gfx950 x2 correction), or None: the
    counters cannot be read from inside the timed run."""
    if scale != 1.0:
        return None
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'pmc_traffic.json')) as f:
            table = json.load(f)
        prefix = '%s_c5_round' % dtype if c5 else '%s_round' % dtype
        rounds = [k for k in table if k.startswith(prefix) and k[len(prefix):].isdigit()]
        e = table[max(rounds, key=lambda k: int(k[len(prefix):]))] if rounds else None       # the latest committed pass
        return {'bytes': float(e['fetch_bytes_per_launch']) + float(e['write_bytes_per_launch']), 'source': e['source']} if e else None
    except (OSError, ValueError, KeyError):
        return None


PMC_ERRORS = []          # why an in-run counter pass did not deliver (surfaced in the record: `pmc_errors`)


def pmc_traffic_in_run(dtype, workload='c3', steps=2, timeout=420):
    """HBM bytes per contraction launch measured IN THIS RUN: two children of this very script under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... WRITE_SIZE` (separate passes, as MI355X_MICROARCH.md prescribes;
    counters only, no API / memory-copy tracing), `steps` + 1 iterations each; FETCH_SIZE is in KiB and counts 64 B per 128-B
    request of a wide streaming read on gfx950 (x2), WRITE_SIZE is taken as reported.  Returns {'bytes', 'source', ...} like
    measured_traffic, or None (no rocprofv3 on PATH, SKF_BENCH_PMC=0, a child failed or timed out): the committed pass then
    stands in, labelled as such."""
    import glob
    import re
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    if dtype != 'bf16' or os.environ.get('SKF_BENCH_PMC', '1') == '0' or not shutil.which('rocprofv3'):
        return None
    sums, calls = {}, {}
    t0 = time.perf_counter()
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        out = tempfile.mkdtemp(prefix='skf_pmc_')
        cmd = ['rocprofv3', '--kernel-trace', '--pmc', counter, '-d', out, '-o', 'pmc', '--', sys.executable,
               os.path.abspath(__file__), '--steps', str(steps), '--warmup', '1', '--workload', workload, '--dtype', dtype,
               '--no-cpu-baseline', '--no-engines', '--no-workloads', '--no-pmc', '--sustained-steps', '0']
        try:
            subprocess.run(cmd, cwd=tempfile.gettempdir(), env=dict(os.environ, TMPDIR=tempfile.gettempdir()),
                           capture_output=True, text=True, timeout=timeout, check=True)
            tot = n = 0.0
            for db in glob.glob(os.path.join(out, '**', '*.db'), recursive=True):
                cur = sqlite3.connect(db).cursor()
                for name, cnt, val in cur.execute("select kernel_name, count(*), sum(value) from counters_collection "
                                                  "where counter_name = ? group by kernel_name", (counter,)):
                    short = re.sub(r'\(.*$', '', name).replace('skf::', '').replace('void ', '')
                    m = re.match(r'gemm_bf16_v2_kernel<\s*\d+\s*,\s*(\d+)\s*,', short)
                    hit = bool(m and m.group(1) == '1')         # TAG = 1: the launches that walk a relation (P, Q)
                    if workload == 'c5':                        # ... and the passes over lists (known entries, sparse 0/1 relations)
                        hit = hit or short.startswith(('srp_bf16', 'srp_vec', 'srp_any', 'binary_spmm'))
                    if hit:
                        tot += float(val)
                        n += cnt
            if n <= 0:
                PMC_ERRORS.append('%s %s: no matching launches in the counter database' % (workload, counter))
                return None
            sums[counter], calls[counter] = tot * 1024.0 * (2.0 if counter == 'FETCH_SIZE' else 1.0), n
        except Exception as exc:
            tail = (getattr(exc, 'stderr', None) or '')[-300:] if hasattr(exc, 'stderr') else ''
            PMC_ERRORS.append('%s %s: %s %s' % (workload, counter, str(exc)[:200], tail))
            return None
        finally:
            shutil.rmtree(out, ignore_errors=True)
    fetch, write = sums['FETCH_SIZE'] / calls['FETCH_SIZE'], sums['WRITE_SIZE'] / calls['WRITE_SIZE']
    return {'bytes': fetch + write, 'fetch_bytes_per_launch': fetch, 'write_bytes_per_launch': write,
            'launches_counted': int(calls['FETCH_SIZE']), 'seconds': time.perf_counter() - t0, 'in_run': True,
            'source': 'this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two child passes of bench.py, %d + 1 '
                      'iterations each), FETCH_SIZE x2 gfx950 correction, mean over the %d %s launches'
                      % (steps, int(calls['FETCH_SIZE']), 'contraction-class (relation / list-pass)' if workload == 'c5' else 'relation-contraction')}


MFMA_PMC_COUNTERS = ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_INSTS_VALU_MFMA_MOPS_BF16', 'SQ_WAIT_ANY',
                     'SQ_WAVE_CYCLES', 'GRBM_GUI_ACTIVE')     # 5 SQ slots of 8, 1 GRBM slot of 2: one pass


def pmc_mfma_in_run(dtype, steps=2, timeout=420):
    """Matrix-core utilisation of the relation contractions measured IN THIS RUN (north_star: "rocprof MFMA utilisation"):
    one more child of this script under `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES ...
    GRBM_GUI_ACTIVE` (counters only), `steps` + 1 iterations of config 3.  Per contraction launch:
        cycles        = GRBM_GUI_ACTIVE / 8 XCDs                      (the counter is summed over the XCDs)
        mfma_busy     = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs)
        clock_ghz     = cycles / the launch's duration in the same trace
        mfma_tflops   = mfma_busy x 2.5 PF x clock / 2.4 GHz           (what the busy share delivers at that clock)
    Returns the record (+ the raw per-launch counter means) or None (no rocprofv3, SKF_BENCH_PMC=0, a failed child)."""
    import glob
    import re
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    if dtype != 'bf16' or os.environ.get('SKF_BENCH_PMC', '1') == '0' or not shutil.which('rocprofv3'):
        return None
    out = tempfile.mkdtemp(prefix='skf_pmc_')
    cmd = ['rocprofv3', '--kernel-trace', '--pmc'] + list(MFMA_PMC_COUNTERS) + ['-d', out, '-o', 'pmc', '--', sys.executable,
           os.path.abspath(__file__), '--steps', str(steps), '--warmup', '1', '--workload', 'c3', '--dtype', dtype,
           '--no-cpu-baseline', '--no-engines', '--no-workloads', '--no-pmc', '--sustained-steps', '0']
    t0 = time.perf_counter()
    try:
        subprocess.run(cmd, cwd=tempfile.gettempdir(), env=dict(os.environ, TMPDIR=tempfile.gettempdir()),
                       capture_output=True, text=True, timeout=timeout, check=True)
        sums, per_kernel, n_launch, dur_ns = {}, {}, 0, 0.0
        is_hit = lambda name: bool(re.match(r'gemm_bf16_v2_kernel<\s*\d+\s*,\s*1\s*,',          # noqa: E731  (TAG = 1: P, Q)
                                            re.sub(r'\(.*$', '', name).replace('skf::', '').replace('void ', '')))
        for db in glob.glob(os.path.join(out, '**', '*.db'), recursive=True):
            cur = sqlite3.connect(db).cursor()
            for name, counter, cnt, val in cur.execute("select kernel_name, counter_name, count(*), sum(value) from "
                                                       "counters_collection group by kernel_name, counter_name"):
                if is_hit(name):
                    sums[counter] = sums.get(counter, 0.0) + float(val)
                    short = re.sub(r'\(.*$', '', name).replace('skf::', '').replace('void ', '')
                    per_kernel.setdefault(short, {})[counter] = (float(val), int(cnt))
            cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
            namecol = 'name' if 'name' in cols else 'kernel_name'
            for name, cnt, tot in cur.execute("select %s, count(*), sum(end - start) from kernels group by %s" % (namecol, namecol)):
                if is_hit(name):
                    n_launch += int(cnt)
                    dur_ns += float(tot)
        if n_launch <= 0 or not sums.get('GRBM_GUI_ACTIVE') or not sums.get('SQ_VALU_MFMA_BUSY_CYCLES'):
            PMC_ERRORS.append('c3 mfma counters: no matching launches in the counter database')
            return None
        cycles = sums['GRBM_GUI_ACTIVE'] / 8.0                     # all launches, per XCD
        busy = sums['SQ_VALU_MFMA_BUSY_CYCLES'] / (cycles * 1024.0)
        clock = cycles / dur_ns if dur_ns > 0 else None            # cycles per ns = GHz
        rec = {'mfma_busy': busy, 'clock_ghz': clock, 'launches_counted': n_launch,
               'avg_launch_ms_under_counters': dur_ns / n_launch / 1e6,
               'mfma_tflops_at_clock': busy * PEAK_TFLOPS['bf16'] * (clock or 0.0) / 2.4,
               'waves_parked': (sums.get('SQ_WAIT_ANY', 0.0) / sums['SQ_WAVE_CYCLES']) if sums.get('SQ_WAVE_CYCLES') else None,
               'per_launch': {k: v / n_launch for k, v in sorted(sums.items())},
               'per_kernel': {k: {c: {'per_launch': v / max(m, 1), 'launches': m} for c, (v, m) in sorted(d.items())}
                              for k, d in sorted(per_kernel.items())},
               'seconds': time.perf_counter() - t0,
               'source': 'this run: rocprofv3 --kernel-trace --pmc %s (one child pass of bench.py, %d + 1 iterations); '
                         'busy = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), clock = cycles / launch time in the '
                         'same trace' % (' '.join(MFMA_PMC_COUNTERS), steps)}
        return rec
    except Exception as exc:
        tail = (getattr(exc, 'stderr', None) or '')[-300:] if hasattr(exc, 'stderr') else ''
        PMC_ERRORS.append('c3 mfma counters: %s %s' % (str(exc)[:200], tail))
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def roofline_record(dtype, n, ranks, spec, k_ms, k_launches, k_flops, steps, elapsed, pmc=None, k_bytes=None,
                    executed=False):
    """Roofline of the launches that walk a relation (hipEvent time `k_ms` over `k_launches` launches).
    executed == False (config 3): SURVEY.md 8(d) algorithmic work -- ONE read of every relation, sum 2 n_i n_j (c_i + c_j)
    flops -- against the launch time.  executed == True (config 5): the flops the launches EXECUTE and the relation bytes they
    read as STORED (bitmaps, index lists, known-entry lists), both counted by the library (skf_plan_get_profile).
    frac = t_min / t_kernel with t_min = max(flops / matrix-core peak, bytes / 8 TB/s); `bound` names the binding term and
    achieved / peak are quoted on it (so frac == achieved / peak)."""
    esz = {'bf16': 2, 'f32': 4, 'f64': 8}[dtype]
    msz = 4 if dtype != 'f64' else 8                       # master type of P / Q
    gsz = 2 if dtype == 'bf16' else msz                    # factor operand of the contraction
    per_iter = sum(3 if m else 2 for _, _, m in spec)      # (dense model) a masked relation recomputes P after completion
    alg_bytes_iter = sum(float(n[i]) * n[j] for i, j, _ in spec) * esz            # ONE read of every relation
    sched_iter = 0.0                                                               # what the launches move from HBM
    for i, j, m in spec:
        rel = float(n[i]) * n[j] * esz
        p_launch = rel + 8.0 * n[j] * ranks[j] * gsz + float(n[i]) * ranks[j] * msz   # P: R + G_j^T per XCD + P
        q_launch = rel + 8.0 * n[i] * ranks[i] * gsz + float(n[j]) * ranks[i] * msz   # Q: R + G_i^T per XCD + Q
        sched_iter += (2 if m else 1) * p_launch + q_launch
    flops_iter = alg_flops(n, spec, ranks)
    peak_tf = PEAK_TFLOPS[dtype]
    rec = {'kernel': 'relation contractions P=R*G_j, Q=R^T*G_i (%s)'
                     % ('gemm_bf16_v2_kernel<BN,TAG=1,AT>' if dtype == 'bf16' else 'gemm_mfma_kernel<..,TAG=1>'),
           'launches': int(k_launches), 'avg_launch_ms': k_ms / k_launches if k_launches else None}
    if executed:
        rec['kernel'] = 'launches that walk a relation: bitmap / dense contractions, 0/1 gathers, known-entry list passes'
        rec['accounting'] = 'executed flops, relation bytes as stored (skf_plan_get_profile)'
    else:
        rec.update({'launches_per_iter': per_iter, 'alg_bytes_per_iter': alg_bytes_iter, 'alg_flops_per_iter': flops_iter,
                    'intensity_algorithmic': flops_iter / alg_bytes_iter, 'intensity_scheduled': flops_iter / sched_iter,
                    'ridge': RIDGE[dtype]})
    if not k_ms or not k_launches:
        rec.update({'bound': None, 'achieved': None, 'peak': None, 'unit': None, 'frac': None, 'traffic': None})
        return rec
    sec = k_ms * 1e-3
    if executed:
        flops, nbytes = float(k_flops), float(k_bytes or 0.0)
    else:
        iters = k_launches / float(per_iter)
        flops, nbytes = flops_iter * iters, alg_bytes_iter * iters
    t_flops, t_bytes = flops / (peak_tf * 1e12), nbytes / (HBM_PEAK_GBS * 1e9)
    mfma_tf, alg_gbs = flops / sec / 1e12, nbytes / sec / 1e9
    rec['mfma'] = {'achieved': mfma_tf, 'peak': peak_tf, 'unit': 'TFLOP/s', 'frac': mfma_tf / peak_tf}
    rec['hbm_algorithmic'] = {'achieved': alg_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': alg_gbs / HBM_PEAK_GBS}
    rec['t_min_ms'] = max(t_flops, t_bytes) * 1e3
    rec['t_kernel_ms'] = k_ms
    if t_flops >= t_bytes:
        rec.update({'bound': 'mfma', 'achieved': mfma_tf, 'peak': peak_tf, 'unit': 'TFLOP/s'})
    else:
        rec.update({'bound': 'hbm', 'achieved': alg_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s'})
    rec['frac'] = max(t_flops, t_bytes) / sec
    if executed:
        rec['traffic'] = pmc['bytes'] if pmc else None
        if pmc:
            rec['traffic_kind'] = ('PMC, ' if pmc.get('in_run') else 'PMC, committed pass: ') + pmc['source']
            rec['executed_bytes_per_launch'] = nbytes / k_launches
        return rec
    sched_gbs = sched_iter * iters / sec / 1e9
    rec['hbm_scheduled'] = {'achieved': sched_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': sched_gbs / HBM_PEAK_GBS}
    rec['traffic_scheduled'] = sched_iter / per_iter        # relation once per contraction + G^T once per XCD + output
    if pmc:
        rec['traffic'] = pmc['bytes']
        rec['traffic_kind'] = ('PMC, ' if pmc.get('in_run') else 'PMC, committed pass: ') + pmc['source']
    else:
        rec['traffic'] = None
        rec['traffic_kind'] = 'no counter pass committed for this workload / size (see traffic_scheduled)'
    rec['traffic_ratio'] = (pmc['bytes'] if pmc else sched_iter / per_iter) / (alg_bytes_iter / per_iter)
    rec['alg_bytes_per_launch'] = alg_bytes_iter / per_iter
    rec['whole_iteration'] = {'mfma_frac': flops_iter * steps / elapsed / 1e12 / peak_tf,
                              'hbm_algorithmic_frac': alg_bytes_iter * steps / elapsed / 1e9 / HBM_PEAK_GBS,
                              'hbm_scheduled_frac': sched_iter * steps / elapsed / 1e9 / HBM_PEAK_GBS}
    return rec


def host_info():
    try:
        ram = os.sysconf('SC_PHYS_PAGES') * os.sysconf('SC_PAGE_SIZE') / 2.0 ** 30
    except (ValueError, OSError, AttributeError):
        ram = None
    return {'cpu_count': os.cpu_count(), 'cores_physical': physical_cores(), 'ram_gib': ram}


def physical_cores():
    """Physical cores from /proc/cpuinfo ((physical id, core id) pairs); None when it cannot be read."""
    try:
        seen, phys, core = set(), None, None
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('physical id'):
                    phys = line.split(':')[1].strip()
                elif line.startswith('core id'):
                    core = line.split(':')[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        seen.add((phys, core))
                    phys = core = None
        return len(seen) or None
    except OSError:
        return None


def blas_threads():
    try:
        from threadpoolctl import threadpool_info
        return int(max([p.get('num_threads', 1) for p in threadpool_info()] or [1]))
    except Exception:
        return int(os.cpu_count() or 1)


def _parallel_hash_uniform(seed, rows, cols, threads=64):
    """fp64 matrix of the counter-based uniforms the device generates (oracle.hash_uniform), filled in row blocks by
    `threads` host threads (NumPy releases the GIL inside the element-wise kernels)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import dfmf_oracle as orc
    out = np.empty((rows, cols), dtype=np.float64)
    chunk = max(1, (1 << 22) // max(cols, 1))                # ~4 M elements per call: small temporaries

    def fill(a):
        b = min(a + chunk, rows)
        out[a:b] = orc.hash_uniform(seed, a * cols, (b - a) * cols).reshape(b - a, cols)
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(fill, range(0, rows, chunk)))
    return out


PARITY_ROWS = 64               # rows of every factor kept for the full-size parity record
PARITY_CHECKPOINTS = (2, 5)    # iterations after which the engine is compared with the oracle (backbones, factor rows)
PARITY_ITERS = PARITY_CHECKPOINTS[-1]          # ... and the relation errors after the last of them
# S = K_i W K_j: a relative perturbation eps of W = G_i^T R G_j reaches S multiplied by up to cond(Gram_i) cond(Gram_j), so
# the backbones are gated at  ||S - S_oracle|| / ||S_oracle||  <=  cond_i cond_j eps(engine)  with the condition numbers of
# the Gram matrices S was formed from (computed from the engine's OWN factors) and eps = at most TEN TIMES the ratio the
# full-size record itself measured on the MI355X (BENCH_r05: f64 1.63e-16, f32 2.54e-9, bf16 1.52e-8 -- rounds 4 / 5 had
# priced eps from the error of one row of P instead and stood 28x / 13x above the f32 / bf16 measurements)
PARITY_S_EPS = {'f64': 1.6e-15, 'f32': 2.5e-8, 'bf16': 1.5e-7}


def parity_rows(n_t):
    """The PARITY_ROWS row indices of a factor of n_t rows that the parity record compares (evenly spread, first and last)."""
    return np.unique(np.linspace(0, n_t - 1, PARITY_ROWS).astype(np.int64))


def _oracle_timing(scale, iters, parallel_data=False, keep=None):
    """Seconds of each of `iters` oracle iterations (NumPy, reference operation order, fp64) on the config-3 graph.
    keep (a path): after every iteration of PARITY_CHECKPOINTS the oracle's backbones and PARITY_ROWS rows of every factor, and
    after the last iteration the three relation errors ||R - G_i S G_j^T||_F (_dfmf.py:306-316: factors AFTER the last
    update, backbones from before it) go to an .npz there -- what the engine's own first `iters` iterations are compared with
    (`parity_full_size`)."""
    from oracle import dfmf_oracle as orc
    n = sizes(scale)
    fill = _parallel_hash_uniform if parallel_data else orc.hash_uniform_matrix
    R = {(i, j): [fill(s, n[i], n[j])] for i, j, s in PAIRS}
    G = {(t, t): orc.hash_uniform_matrix(100 + k, n[t], RANKS[t]) for k, t in enumerate(TYPES)}
    times, out = [], {'iters': iters, 'checkpoints': np.array([c for c in PARITY_CHECKPOINTS if c <= iters] or [iters])}
    for it in range(1, iters + 1):
        t0 = time.perf_counter()
        S, _ = orc._update_S(R, G)
        G = orc._update_G(R, G, S, {}, {}, True)
        times.append(time.perf_counter() - t0)
        if keep and it in out['checkpoints']:
            for i, j, _ in PAIRS:
                out['S_%s_%s@%d' % (i, j, it)] = S[i, j][0]
            for t in TYPES:
                out['G_%s@%d' % (t, it)] = G[t, t][parity_rows(n[t])]
    if keep:
        t0 = time.perf_counter()
        errs = orc.relation_errors_blocked(R, G, S)
        out['err_seconds'] = time.perf_counter() - t0
        for i, j, _ in PAIRS:
            out['err_%s_%s' % (i, j)] = errs[i, j][0]
        np.savez(keep, **out)
    return times, n


def _full_size_child(keep=None):
    """`python bench.py --cpu-full-child [--parity-out file]`: PARITY_ITERS oracle iterations at full size (the last one is
    reported: BLAS threads and pages warm) after a 1/10-scale warm-up; prints a JSON line.  Runs in a child process so that
    the parent can bound it in time and memory."""
    _oracle_timing(0.1, 1)
    t0 = time.perf_counter()
    times, n = _oracle_timing(1.0, PARITY_ITERS, parallel_data=True, keep=keep)
    print(json.dumps({'times': times, 'total_seconds': time.perf_counter() - t0}))


def gram_condition(G):
    """cond_2 of G^T G (f64, host) -- of the Gram matrix a backbone is formed from."""
    g = np.asarray(G, dtype=np.float64)
    lam = np.linalg.eigvalsh(g.T.dot(g))
    return float(lam[-1] / max(lam[0], 1e-300))


def parity_record(oracle_npz, engine, dtype=None, s_eps=None):
    """`parity_full_size` of one engine: its first PARITY_ITERS iterations from the hash-generated G0 against the oracle's
    (same inputs, FULL size) -- per checkpoint the max relative deviation of the backbones (Frobenius) and of PARITY_ROWS rows
    of every factor, after the last iteration that of the three relation errors; and the GATE on the backbones: every
    relation's deviation over cond(Gram_i) cond(Gram_j) (the engine's own factors) against PARITY_S_EPS[dtype].
    north_star: results must match the reference path on the same inputs."""
    z = np.load(oracle_npz)
    rel = lambda a, b: float(np.linalg.norm(np.asarray(a, dtype=np.float64) - b) / max(np.linalg.norm(b), 1e-300))   # noqa: E731
    cps = [int(c) for c in z['checkpoints']]
    rec = {'iters': int(z['iters']), 'checkpoints': {}}
    worst_s = worst_g = worst_gate = 0.0
    for c in cps:
        s_dev = {(i, j): rel(engine['S_%s_%s@%d' % (i, j, c)], z['S_%s_%s@%d' % (i, j, c)]) for i, j, _ in PAIRS}
        g_dev = max(rel(engine['G_%s@%d' % (t, c)], z['G_%s@%d' % (t, c)]) for t in TYPES)
        cp = {'S_relerr': max(s_dev.values()), 'G_rows_relerr': g_dev}
        if ('kappa_%s@%d' % (TYPES[0], c)) in engine:
            kap = {t: float(engine['kappa_%s@%d' % (t, c)]) for t in TYPES}
            cp['gram_condition'] = kap
            cp['S_relerr_over_conditioning'] = max(v / (kap[i] * kap[j]) for (i, j), v in s_dev.items())
            worst_gate = max(worst_gate, cp['S_relerr_over_conditioning'])
        rec['checkpoints'][str(c)] = cp
        worst_s, worst_g = max(worst_s, cp['S_relerr']), max(worst_g, g_dev)
    rec.update({'S_relerr': worst_s, 'G_rows_relerr': worst_g,
                'err_relerr': max(abs(float(engine['err_%s_%s' % (i, j)]) / float(z['err_%s_%s' % (i, j)]) - 1.0) for i, j, _ in PAIRS),
                'oracle_err': {'%s-%s' % (i, j): float(z['err_%s_%s' % (i, j)]) for i, j, _ in PAIRS}})
    eps = s_eps if s_eps is not None else PARITY_S_EPS.get(dtype)          # (s_eps: tests at another scale bring their own)
    if eps is not None and worst_gate > 0.0:
        rec['S_gate'] = {'S_relerr_over_conditioning': worst_gate, 'eps': eps, 'ok': bool(worst_gate <= eps)}
    return rec


def cpu_baseline_start(full='auto', parity_out=None):
    """Start the full-size oracle child of `cpu_baseline` in the background (Popen) when this host qualifies, so that the
    GPU-only sub-records of the default run proceed beside it (the child runs 128 BLAS threads of the host's 256 logical
    cores, the parent one or two that enqueue launches); None otherwise.  `cpu_baseline(..., child=...)` collects it."""
    host = host_info()
    want_full = full is True or (full == 'auto' and (host['ram_gib'] or 0) >= 256 and (host['cpu_count'] or 0) >= 32)
    if not want_full:
        return None
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-full-child']
    if parity_out:
        cmd += ['--parity-out', parity_out]
    try:
        return {'proc': subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True), 't0': time.perf_counter()}
    except Exception:
        return None


def cpu_baseline(full='auto', parity_out=None, child=None):
    """Oracle (kind=port) on the host cores.  When the host can hold the fp64 graph (88 GB + temporaries: RAM >= 256 GiB
    and >= 32 cores) PARITY_ITERS iterations are timed at FULL size in a child process bounded to 420 s and the
    second is reported; otherwise, or when that fails, the 1/10-linear-scale sample is timed and scaled by the n_i*n_j work
    ratio."""
    host = host_info()
    base = {'unit': 'iters/s', 'cores': blas_threads(), 'cores_physical': host['cores_physical'], 'kind': 'port',
            'host_cpu_count': host['cpu_count'], 'host_ram_gib': host['ram_gib']}
    want_full = full is True or (full == 'auto' and (host['ram_gib'] or 0) >= 256 and (host['cpu_count'] or 0) >= 32)
    note = ''
    if want_full:
        import subprocess
        try:
            if child is not None:                  # started earlier, beside the GPU sub-records
                left = max(420.0 - (time.perf_counter() - child['t0']), 1.0)
                try:
                    stdout, _ = child['proc'].communicate(timeout=left)
                except subprocess.TimeoutExpired:
                    child['proc'].kill()
                    raise
                beside = ' (run beside the GPU-only sub-records of this bench: one or two busy host threads)'
            else:
                cmd = [sys.executable, os.path.abspath(__file__), '--cpu-full-child']
                if parity_out:
                    cmd += ['--parity-out', parity_out]
                stdout = subprocess.run(cmd, capture_output=True, text=True, timeout=420).stdout
                beside = ''
            r = json.loads([l for l in stdout.splitlines() if l.startswith('{')][-1])
            base.update({'value': 1.0 / r['times'][-1], 'projection': False,
                         'sample': 'oracle (NumPy fp64, reference op order, 3 big GEMMs per relation, scipy pinv) at FULL size, '
                                   'device-identical inputs: iterations %s s, the last one reported (%.0f s with the 88 GB fill)%s'
                                   % ('/'.join('%.1f' % t for t in r['times']), r['total_seconds'], beside)})
            return base
        except Exception as exc:                   # time-out, memory, a host without the packages ...
            note = ' (full-size run failed: %s)' % (str(exc)[:120],)
    _oracle_timing(0.1, 1)
    times, n = _oracle_timing(0.1, 3)
    sample_ips = 1.0 / min(times)
    ratio = alg_flops(n) / alg_flops(FULL)       # the n_i*n_j work shrinks by 100
    base.update({'value': sample_ips * ratio, 'projection': True,
                 'sample': 'oracle at 1/10 linear scale (%dx%d / %dx%d / %dx%d): %.3f it/s measured, scaled by the work '
                           'ratio %.4f%s' % (n['t1'], n['t2'], n['t1'], n['t3'], n['t2'], n['t3'], sample_ips, ratio, note)})
    return base


def _tests_path():
    p = os.path.join(ROOT, 'tests')
    if p not in sys.path:
        sys.path.insert(0, p)


def cpu_baseline_c5(scale=0.25, target=1.0):
    """Oracle dfmc (reference operation order incl. the boolean-mask completion, _dfmc.py:319-325) on the MovieLens-style
    graph at `scale` of the linear sizes (constraints on Movie only, as in the device workload), projected to `target` by
    the cell ratio (scale / target)^2 (labelled)."""
    _tests_path()
    from helpers import movielens_style_graph
    from oracle import dfmf_oracle as orc
    n = sizes(scale, C5_FULL)
    R, M, Theta, types, ranks = movielens_style_graph(n, C5_RANKS)
    Theta.pop(('user', 'user'), None)
    G0 = {(t, t): orc.hash_uniform_matrix(100 + k, n[t], C5_RANKS[t]) for k, t in enumerate(types)}
    t0 = time.perf_counter()
    orc.dfmc(R, M, Theta, types, ranks, max_iter=1, G0=G0)
    t1 = time.perf_counter()
    orc.dfmc(R, M, Theta, types, ranks, max_iter=3, G0=G0)
    t3 = time.perf_counter()
    per_iter = max(((t3 - t1) - (t1 - t0)) / 2.0, 1e-9)
    ratio = (scale / target) ** 2
    return {'value': ratio / per_iter, 'unit': 'iters/s', 'cores': blas_threads(), 'cores_physical': physical_cores(),
            'kind': 'port', 'projection': True,
            'sample': 'oracle dfmc at %.3f linear scale (%d users x %d movies): %.3f s per iteration, projected by the cell '
                      'ratio %.4f (measured once at FULL size, `--workload c5 --cpu-baseline full`: 26.4 s per iteration = 0.0379 '
                      'it/s on 128 cores, profiles/r04_c5_cpu_full.txt -- the projection is 1.9x pessimistic)'
                      % (scale, n['user'], n['movie'], per_iter, ratio)}


def _c5_full_child(scale=1.0):
    """`python bench.py --cpu-c5-child [--scale x]`: the oracle's dfmc (reference operation order, boolean-mask completion)
    on the MovieLens-style graph at `scale` (1.0 = FULL size: 100k users x 40k movies), data of the same distributions as the
    device workload drawn by 64 host threads; two iterations timed through the callback, the second one reported.  A child
    process, so that the parent can bound it in time and memory (R, its copy and the 32 GB reconstruction: ~150 GB)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import dfmf_oracle as orc
    n = sizes(scale, C5_FULL)

    def uniform(seed, rows, cols):
        out = np.empty((rows, cols))
        chunk = max(1, (1 << 23) // max(cols, 1))

        def fill(a):
            b = min(a + chunk, rows)
            out[a:b] = np.random.default_rng([seed, a]).random((b - a, cols))
        with ThreadPoolExecutor(64) as ex:
            list(ex.map(fill, range(0, rows, chunk)))
        return out
    t_fill = time.perf_counter()
    R, M = {}, {}
    for i, j, seed, dens in C5_PAIRS:
        u = uniform(seed, n[i], n[j])
        if dens is None:
            np.multiply(u, 10.0, out=u)
            np.floor(u, out=u)
            u += 1.0
            u /= 10.0
            R[i, j] = [u]
            M[i, j] = [uniform(seed + 100, n[i], n[j]) < 0.98]
        else:
            R[i, j] = [(u < dens).astype(np.float64)]
            M[i, j] = [None]
    nm = n['movie']
    sim = uniform(60, nm, nm) < 1.0 / nm
    sim = -0.001 * (sim | sim.T)
    np.fill_diagonal(sim, 0.0)
    Theta = {('movie', 'movie'): [0.01 * np.eye(nm), sim]}
    G0 = {(t, t): orc.hash_uniform_matrix(100 + k, n[t], C5_RANKS[t]) for k, t in enumerate(C5_TYPES)}
    t_fill = time.perf_counter() - t_fill
    stamps = [time.perf_counter()]
    orc.dfmc(R, M, Theta, C5_TYPES, C5_RANKS, max_iter=2, G0=G0, callback=lambda g, s_, it: stamps.append(time.perf_counter()))
    print(json.dumps({'times': [b - a for a, b in zip(stamps, stamps[1:])], 'fill_seconds': t_fill,
                      'users': n['user'], 'movies': n['movie']}))


def cpu_baseline_c5_full(scale=1.0, timeout=900):
    """The config-5 oracle at FULL size in a bounded child (`--workload c5 --cpu-baseline full`; not part of the default
    run: it is minutes of CPU work): {'value', 'projection': False, ...} or None."""
    import subprocess
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-c5-child', '--scale', str(scale)],
                             capture_output=True, text=True, timeout=timeout)
        r = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    except Exception:
        return None
    return {'value': 1.0 / r['times'][-1], 'unit': 'iters/s', 'cores': blas_threads(), 'cores_physical': physical_cores(),
            'kind': 'port', 'projection': False,
            'sample': 'oracle dfmc (NumPy fp64, reference op order, boolean-mask completion) at FULL size (%d users x %d movies): '
                      'iterations %s s, the last one reported (%.0f s of data fill)'
                      % (r['users'], r['movies'], '/'.join('%.1f' % t for t in r['times']), r['fill_seconds'])}


def cpu_baseline_c5_half(timeout=300):
    """The config-5 oracle leg of the default run: MEASURED (not projected) at 1/2 linear scale -- 50k users x 20k movies, a
    quarter of the cells, where OpenBLAS is still efficient (the 1/4-scale projection of rounds 3-4 was 1.9x pessimistic) --
    in the bounded child of `--cpu-baseline full`; `value` is the rate AT THAT SCALE (`scale: 0.5` says so),
    `value_times_cell_ratio` the same number times 1/4 for orientation against the full-size GPU rate, and the one full-size
    measurement of round 4 (0.0379 it/s, profiles/r04_c5_cpu_full.txt) is quoted."""
    r = cpu_baseline_c5_full(0.5, timeout)
    if r is None:
        return None
    r.update({'scale': 0.5, 'value_times_cell_ratio': r['value'] * 0.25, 'full_size_measured_in_round_4': 0.0379,
              'sample': r['sample'].replace('at FULL size', 'at 1/2 linear scale')})
    return r


def mid_size_record(dtype='bf16', scale=0.1):
    """`workloads.c3_tenth`: SURVEY's 1/10-scale probe graph (5k x 10k / 5k x 4k / 10k x 4k, ranks 128/256/256; the reference
    took 1.4-2.4 s per iteration there) -- between the three-launch schedule of small graphs (ranks <= 64) and the
    sizes the relation pipeline is tuned for: it/s of the three engines, launches per iteration, and the oracle's rate on this
    host beside them."""
    out = {'config': 'BASELINE configs[2] at 1/10 linear scale: 5000x10000 / 5000x4000 / 10000x4000, ranks 128/256/256', 'scale': scale}
    for dt in (dtype, 'f32', 'f64'):
        try:
            w = run_workload('c3', dt, 50, 5, scale=scale)
            out[dt] = {'value': 50 / w['elapsed'], 'unit': 'iters/s', 'ms_per_step': w['elapsed'] / 50 * 1e3,
                       'launches_per_step': w['launches_per_step'], 'host_enqueue_ms_per_step': w['enqueue_ms_per_step']}
        except Exception as exc:
            out[dt] = {'error': str(exc)[:200]}
    try:
        _oracle_timing(scale, 1)
        times, _ = _oracle_timing(scale, 3)
        out['oracle'] = {'value': 1.0 / min(times), 'unit': 'iters/s', 'cores': blas_threads(), 'kind': 'port',
                         'sample': 'NumPy fp64, reference op order, best of 3 iterations'}
    except Exception as exc:
        out['oracle'] = {'error': str(exc)[:200]}
    return out


def bench_dicty(iters=100):
    """BASELINE configs[1]: the dicty graph (ann 1219 x 116, expr 1219 x 282, Theta = ppi; ranks 50/15/5), Dfmf from the
    golden G0 -- f32 and f64 engines and the NumPy oracle on the same host."""
    import torch
    _tests_path()
    from helpers import golden, dicty_graph, g0_from
    from oracle import dfmf_oracle as orc
    import skfusion_amd._native as nat
    from skfusion_amd._engine import DevicePlan, flatten_relations, flatten_thetas
    z = golden('c2_dicty.npz')
    R, Theta, types, rank = dicty_graph()
    G0 = g0_from(z, 'dfmf/', types)
    n = {'gene': R['gene', 'go'][0].shape[0], 'go': R['gene', 'go'][0].shape[1], 'exc': R['gene', 'exc'][0].shape[1]}
    out = {'graph': 'dicty: ann %dx%d, expr %dx%d, ppi constraint, ranks 50/15/5' % (n['gene'], n['go'], n['gene'], n['exc']),
           'iters': iters}
    for dtype in ('f32', 'f64'):
        plan = DevicePlan(types, n, rank, flatten_relations(R), flatten_thetas(Theta), nat.SKF_DFMF, dtype=dtype)
        for t in types:
            plan.set_factor(t, G0[t, t])
        plan.iterate(5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan.iterate(iters)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        err = float(np.sqrt(plan.relation_sqerr(0)))
        out[dtype] = {'value': iters / dt, 'unit': 'iters/s', 'ms_per_step': dt / iters * 1e3, 'err_ann': err}
        plan.close()
    orc.dfmf(R, Theta, types, rank, max_iter=2, G0=G0)
    t0 = time.perf_counter()
    orc.dfmf(R, Theta, types, rank, max_iter=20, G0=G0)
    dt = time.perf_counter() - t0
    out['cpu_baseline'] = {'value': 20 / dt, 'unit': 'iters/s', 'cores': blas_threads(), 'cores_physical': physical_cores(),
                           'kind': 'port', 'sample': '20 oracle iterations (NumPy fp64) in %.2f s' % dt}
    return out


def run_workload(workload, dtype, steps, warmup, scale=1.0, data='uniform', mode='restarts', rank=0, world=1, dist=None,
                 backend='nccl', emulate=None, parity=False, sustained=0):
    """One engine on one workload.  Returns dict(elapsed, k_ms, k_launches, k_flops, k_bytes, rmse, n, spec, ranks, types).
    emulate = (k, W): this process computes what rank k of W would in mode 'owned', exchanges skipped (timing only).
    parity: the first PARITY_ITERS iterations are run apart and their backbones, PARITY_ROWS factor rows and relation errors
    kept (`parity` of the result; config 3, one GPU) -- the engine's side of `parity_full_size`; the timed steps follow on."""
    import torch
    import skfusion_amd._native as nat
    from skfusion_amd._engine import DevicePlan, fill_uniform
    c5 = (workload == 'c5')
    types, ranks_ = (C5_TYPES, C5_RANKS) if c5 else (TYPES, RANKS)
    n = sizes(scale, C5_FULL if c5 else FULL)
    if emulate:
        mode, (rank, world) = 'owned', emulate
    sharded = (mode in ('relations', 'rows', 'owned') and world > 1)
    variant = nat.SKF_DFMC if c5 else nat.SKF_DFMF
    esz = {'bf16': 2, 'f32': 4, 'f64': 8}[dtype]
    # the whole graph as (row, col, masked) + a maker of relation k's device matrices
    if c5:
        full_rels, full_thetas = c5_graph(n, dtype)
        spec = [(i, j, dens is None) for i, j, _, dens in C5_PAIRS]

        def make(k):
            return full_rels[k][2], full_rels[k][3]
    else:
        full_rels, full_thetas = None, []
        spec = [(i, j, False) for i, j, _ in PAIRS]
        planted = {}

        def make(k):                   # same values whatever the sharding (counter-based generator)
            return c3_relation(k, n, dtype, data, planted), None
    part_rel = [(i, j, None, None) for i, j, _ in spec]
    part_th = [(t, None) for t, _ in full_thetas]
    local_index = list(range(len(spec)))            # global index of every relation of this plan
    thetas = list(full_thetas)
    if sharded and mode == 'relations':       # this rank keeps only its share of the relations
        from skfusion_amd._distributed import partition_relations
        owner, th_owner = partition_relations(part_rel, part_th, n, ranks_)
        local_index = [k for k, o in enumerate(owner) if o == rank]
        thetas = [t for t, o in zip(full_thetas, th_owner) if o == rank]
    if sharded and mode == 'rows':            # every relation listed, with this rank's row block of it
        from skfusion_amd._distributed import partition_rows
        blocks, th_owner = partition_rows(part_rel, part_th, n, ranks_, align=256 if min(n.values()) >= 4096 else 64)
        thetas = [t for t, o in zip(full_thetas, th_owner) if o == rank]
        rels = []
        for k, ((i, j, masked), blk) in enumerate(zip(spec, blocks)):
            mine = [b for b in blk if b[0] == rank]
            if not mine:
                rels.append((i, j, None, None, dict(absent=True, row_begin=0, n_rows=0, col_side=False, masked=masked)))
                continue
            _, a, cnt = mine[0]
            rdata, mask = make(k)
            rels.append((i, j, rdata.rows(a, cnt, esz), None if mask is None else mask.rows(a, cnt, 1),
                         dict(absent=False, row_begin=a, n_rows=cnt, col_side=(a == 0), masked=masked)))
            del rdata, mask
    elif sharded and mode == 'owned':         # the rows of every type this rank owns, with the matching rows of the relations
        from skfusion_amd._engine import owned_rows
        rels = []
        for k, (i, j, masked) in enumerate(spec):
            a, cnt, _ = owned_rows(dtype, n[i], rank, world)
            if cnt == 0:
                blk = dict(absent=True, row_begin=0, n_rows=0, masked=masked)
                if masked:
                    from skfusion_amd._engine import known_lists_pay
                    blk['known_lists'] = known_lists_pay(make(k)[1].known, n[i], n[j], ranks_[i], dtype)
                rels.append((i, j, None, None, blk))
                continue
            rdata, mask = make(k)
            blk = dict(absent=False, row_begin=a, n_rows=cnt, masked=masked)
            if dtype == 'bf16' and mask is None:
                blk['binary'] = bool(rdata.binary)
            mrows = None
            if mask is not None:
                # lists of the known entries under row ownership: decided on the WHOLE relation, for all ranks alike; the
                # local rows' count bounds this rank's lists
                from skfusion_amd._engine import known_lists_pay
                blk['known_lists'] = known_lists_pay(mask.known, n[i], n[j], ranks_[i], dtype)
                mrows = mask.rows(a, cnt, 1)
                mrows.known = int((mask.buf.owner[a:a + cnt] == 0).sum().item())
            rels.append((i, j, rdata.rows(a, cnt, esz), mrows, blk))
            del rdata, mask
    else:
        rels = [(spec[k][0], spec[k][1]) + make(k) for k in local_index]
    plan = DevicePlan(types, n, ranks_, rels, thetas, variant, dtype=dtype,
                      part=(rank, world) if sharded and mode in ('rows', 'owned') else None,
                      owned=sharded and mode == 'owned')
    if dtype == 'bf16':          # the plan keeps its own padded bf16 copy (or bitmap / lists) of every relation
        plan.release_relation_data()
        rels = full_rels = None
        torch.cuda.empty_cache()
    for k, t in enumerate(types):      # one random restart per rank: G0 seed depends on the rank
        seed = 100 + k + (0 if sharded else 10 * rank)       # sharded: replicated factors
        plan.set_factor(t, fill_uniform((n[t], ranks_[t]), seed, MASTER[dtype]))
    exchange = comm = None
    if emulate:
        plan.attach_null_comm(rank, world)
        exchange = plan.exchange_bytes(world)
    elif sharded:                      # the library issues the exchanges itself: RCCL, or torch.distributed callbacks over gloo
        plan.attach_comm()
        exchange = plan.exchange_bytes(world)
        comm = plan.comm_info()
    step = plan.iterate if not sharded else {'rows': plan.iterate_rows, 'relations': plan.iterate_sharded,
                                             'owned': plan.iterate_dist}[mode]

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    kept = None
    if parity and not sharded and not c5:
        kept, done = {}, 0
        for cp in PARITY_CHECKPOINTS:
            if cp - 1 > done:
                step(cp - 1 - done)
            for t in types:            # the factors the backbones of iteration `cp` are formed from: their conditioning
                kept['kappa_%s@%d' % (t, cp)] = gram_condition(plan.get_factor(t))
            step(1)
            done = cp
            for q, (i, j, _) in enumerate(spec):
                kept['S_%s_%s@%d' % (i, j, cp)] = plan.get_backbone(q)
            for t in types:
                kept['G_%s@%d' % (t, cp)] = plan.get_factor(t)[parity_rows(n[t])]
        for q, (i, j, _) in enumerate(spec):
            kept['err_%s_%s' % (i, j)] = float(np.sqrt(max(plan.relation_sqerr(q), 0.0)))
    if warmup:
        step(warmup)
    sync()
    plan.set_profiling(True)
    from skfusion_amd._engine import launch_count
    l0 = launch_count()
    t0 = time.perf_counter()
    step(steps)
    enqueued = time.perf_counter() - t0        # host time to issue the launches of `steps` iterations (nothing waited for)
    launches = (launch_count() - l0) / float(steps)
    sync()
    elapsed = time.perf_counter() - t0
    k_ms, k_launches, k_flops, k_bytes = plan.get_profile()
    plan.set_profiling(False)
    sus = None
    if sustained and not emulate:              # a second, long timed region on the same plan: the rate the part sustains
        sync()                                 # (the chip clocks down under real operands: 1.67 GHz vs 2.41 GHz on zeros)
        s0 = time.perf_counter()
        step(sustained)
        sync()
        s1 = time.perf_counter() - s0
        sus = {'steps': int(sustained), 'seconds': s1, 'value': sustained / s1, 'unit': 'iters/s', 'ms_per_step': s1 / sustained * 1e3}
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device='cuda' if backend == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    rmse = {}
    if emulate:
        pass                                  # (no exchanges: the factors mean nothing)
    elif sharded and mode in ('rows', 'owned'):   # every rank holds the squared error of its row blocks
        sq = torch.tensor([plan.relation_sqerr(k) for k in range(len(spec))], dtype=torch.float64,
                          device='cuda' if backend == 'nccl' else 'cpu')
        dist.all_reduce(sq)
        for k, (i, j, _) in enumerate(spec):
            rmse['%s-%s' % (i, j)] = float(np.sqrt(float(sq[k]) / (n[i] * n[j])))
    else:
        for q, k in enumerate(local_index):
            i, j, _ = spec[k]
            rmse['%s-%s' % (i, j)] = float(np.sqrt(max(plan.relation_sqerr(q), 0.0) / (n[i] * n[j])))
    plan.close()
    del plan
    torch.cuda.empty_cache()
    return {'elapsed': elapsed, 'k_ms': k_ms, 'k_launches': k_launches, 'k_flops': k_flops, 'k_bytes': k_bytes,
            'rmse': rmse, 'n': n, 'spec': spec, 'ranks': ranks_, 'types': types, 'sharded': sharded,
            'exchange_bytes': exchange, 'parity': kept, 'sustained': sus, 'enqueue_ms_per_step': enqueued / steps * 1e3,
            'launches_per_step': launches, 'comm': comm,
            'quantisation': ({'%s-%s' % (i, j): planted.get('quant_%d' % k) for k, (i, j, _) in enumerate(spec)}
                             if (not c5 and data == 'planted' and dtype == 'bf16') else None)}


XGMI_LINK_GBS = 153.0          # MI355X_MICROARCH.md: 7 xGMI links per GPU, ~153 GB/s each (fully connected 8-GPU node)


def emulated_ranks(args):
    """`--emulate-rank k/W`: per-rank compute time of the ownership-sharded fit measured on THIS GPU (exchanges skipped) and
    the wire time of the rank's exchange bytes modelled on the xGMI links (reduce-scatter / all-gather of a buffer over a
    fully connected node: every peer's share crosses its own link, so a rank's bytes move at up to 7 links x 153 GB/s;
    `wire_ms_ring` prices the same bytes on ONE link, the bound of a ring).  No multi-GPU node was available to the
    builder: the sum is a model, the compute term is a measurement."""
    which, world = args.emulate_rank.split('/')
    world = int(world)
    ks = list(range(world)) if which == 'all' else [int(which)]
    out = {'metric': 'per-rank compute of the ownership-sharded iteration (exchanges skipped)', 'world': world,
           'dtype': args.dtype, 'workload': args.workload, 'scale': args.scale, 'steps': args.steps, 'ranks': []}
    for k in ks:
        w = run_workload(args.workload, args.dtype, args.steps, args.warmup, args.scale, args.data, emulate=(k, world))
        ms = w['elapsed'] / args.steps * 1e3
        xb = float(w['exchange_bytes'])
        out['ranks'].append({'rank': k, 'compute_ms_per_step': ms, 'host_enqueue_ms_per_step': w['enqueue_ms_per_step'],
                             'launches_per_step': w['launches_per_step'],
                             'exchange_bytes_per_step': xb,
                             'wire_ms_all_links': xb / (7 * XGMI_LINK_GBS * 1e9) * 1e3,
                             'wire_ms_ring': xb / (XGMI_LINK_GBS * 1e9) * 1e3})
    worst = max(r['compute_ms_per_step'] for r in out['ranks'])
    out['max_compute_ms_per_step'] = worst
    out['modelled_ms_per_step'] = {'no_overlap_all_links': worst + out['ranks'][0]['wire_ms_all_links'],
                                   'no_overlap_ring': worst + out['ranks'][0]['wire_ms_ring'],
                                   'full_overlap': max(worst, out['ranks'][0]['wire_ms_all_links'])}
    return out


def rank_of_8_record(dtype='bf16', rank=3, world=8, steps=10, warmup=3):
    """`workloads.rank_of_8` of the default one-GPU line: what ONE rank of 8 of the ownership-sharded fit (north_star: "reported
    at 1, 2, 4 and 8 MI355X"; reference per-block tasks _dfmf.py:69-73, _dfmc.py:341-345) computes per iteration, MEASURED on
    this GPU with the exchanges skipped (null communicator: partial sums scaled, factors held at G0), for BASELINE configs[2]
    and configs[4], with the bytes the rank would exchange and their modelled wire time.  The compute term is a measurement;
    the 8-GPU iteration time is a model until a SCALE run's `strong` sub-record replaces it."""
    out = {'what': 'compute of rank %d of %d of the ownership-sharded iteration on this GPU, exchanges skipped (bench.py '
                   '--emulate-rank %d/%d); wire = exchange bytes over 7 xGMI links x %.0f GB/s (all links) or one link (ring)'
                   % (rank, world, rank, world, XGMI_LINK_GBS), 'rank': rank, 'world': world, 'dtype': dtype}
    for key, wl in (('c3', 'c3'), ('c5', 'c5')):
        try:
            w = run_workload(wl, dtype, steps, warmup, emulate=(rank, world))
            ms = w['elapsed'] / steps * 1e3
            xb = float(w['exchange_bytes'])
            out[key] = {'compute_ms_per_step': ms, 'steps': steps, 'warmup': warmup,
                        'launches_per_step': w['launches_per_step'], 'host_enqueue_ms_per_step': w['enqueue_ms_per_step'],
                        'exchange_bytes_per_rank_and_iter': xb,
                        'wire_ms_all_links': xb / (7 * XGMI_LINK_GBS * 1e9) * 1e3, 'wire_ms_ring': xb / (XGMI_LINK_GBS * 1e9) * 1e3,
                        'contraction_ms_per_step': w['k_ms'] / steps}
        except Exception as exc:
            out[key] = {'error': str(exc)[:300]}
    return out


def strong_record(dtype, rank, world, dist, backend, steps=10, warmup=3, scale=1.0):
    """`strong` sub-record of an N > 1 run: after the restarts measurement the SAME process group runs ONE fit sharded by
    ownership (`--mode owned`: reduce-scatter of the partial Q, all-gather of the updated rows, c x c all-reduces -- the
    exchange step of north_star / BASELINE configs[4]) on configs[2] and configs[4], so that one SCALE run yields both curves:
    it/s of the one fit, bytes a rank sends per iteration, and what the transport says about itself (RCCL: ncclCommCount).
    Every rank takes part; rank 0 keeps the record."""
    out = {'mode': 'owned', 'scaling': 'strong', 'n_gpus': world, 'dtype': dtype, 'backend': backend}
    if os.environ.get('SKF_BENCH_DIE_IN_STRONG') == str(rank):       # rehearsal of a rank lost inside this leg (tools/gpu_round2.sh)
        import signal
        os.kill(os.getpid(), signal.SIGKILL)
    for key, wl in (('c3', 'c3'), ('c5', 'c5')):
        try:
            w = run_workload(wl, dtype, steps, warmup, scale, 'uniform', 'owned', rank, world, dist, backend)
            comm = w['comm'] or {}
            out[key] = {'value': steps / w['elapsed'], 'unit': 'iters/s', 'steps': steps, 'warmup': warmup,
                        'ms_per_step': w['elapsed'] / steps * 1e3,
                        'exchange_bytes_per_rank_and_iter': w['exchange_bytes'],
                        'launches_per_step': w['launches_per_step'],
                        'transport': comm.get('transport'), 'transport_ranks': comm.get('transport_ranks'),
                        'rmse': w['rmse']}
        except Exception as exc:           # (a rank that fails here fails on every rank: same graph, same plan)
            out[key] = {'error': str(exc)[:300]}
    return out


def run_bounded(fn, seconds, device=None):
    """fn() in a worker thread with a time limit: (result, timed_out).  The `strong` leg of an N > 1 run enters collectives
    that no single-GPU box can rehearse with more than one real rank; should they hang, the measured restarts line must
    still be printed -- the caller then prints it and leaves through os._exit, skipping the process-group teardown."""
    import threading
    box = {}

    def work():
        try:
            if device is not None:
                import torch
                torch.cuda.set_device(device)          # (the current device is per thread)
            box['result'] = fn()
        except Exception as exc:
            box['result'] = {'error': str(exc)[:300]}
    th = threading.Thread(target=work, name='bench-strong', daemon=True)
    th.start()
    th.join(seconds)
    if th.is_alive():
        return {'error': 'timed out after %.0f s (SKF_STRONG_TIMEOUT)' % seconds}, True
    return box.get('result'), False


class LineWatchdog:
    """Keeps the measured line of an N > 1 run safe while rank 0 enters the `strong` leg -- collectives over RCCL that no
    single-GPU box can rehearse with more than one real rank.  A forked child (it touches no GPU state: two system calls)
    holds the line as it stands and waits on a pipe: if rank 0 dies in there (a fault inside a collective takes the whole
    process), the pipe closes and the child prints the line; if rank 0 comes back, it disarms the child and prints the full
    line itself.  Hangs are bounded by run_bounded, a TERM from the launcher (a PEER died) by the signal handler of main()."""

    def __init__(self, line):
        import signal
        self.w = self.pid = None
        try:
            r, w = os.pipe()
            sys.stdout.flush()
            pid = os.fork()
        except OSError:
            return
        if pid == 0:
            try:
                os.close(w)
                for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
                    signal.signal(sig, signal.SIG_IGN)
                if os.read(r, 1) != b'd':
                    os.write(1, line.encode() + b'\n')
            finally:
                os._exit(0)
        os.close(r)
        self.w, self.pid = w, pid

    def disarm(self):
        if self.w is None:
            return
        try:
            os.write(self.w, b'd')
            os.close(self.w)
            os.waitpid(self.pid, 0)
        except OSError:
            pass
        self.w = None


def compact_roofline(r):
    return {k: r.get(k) for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 't_min_ms', 't_kernel_ms', 'launches',
                                  'accounting', 'traffic', 'executed_bytes_per_launch', 'traffic_kind') if k in r}


def other_workloads(dtype='bf16', dicty=None):
    """The `workloads` sub-record of the default run (single GPU) -- GPU legs only: BASELINE configs[4], configs[1] (`dicty`:
    measured by the caller while the host was still quiet -- that leg is bound by the host's launch rate), the planted-data
    RMSE of configs[2], the rank-of-8 emulation and the 1/10-scale graph; every leg reports its own failure instead of sinking
    the headline line.  The host legs (the oracle timing of configs[4], its PMC children) follow in `other_workloads_host`,
    after the full-size oracle child of the headline has finished: CPU timings are not taken beside one another."""
    out = {'c2_dicty': dicty if dicty is not None else dicty_record()}
    try:
        w = run_workload('c5', dtype, 10, 3)
        roof = roofline_record(dtype, w['n'], w['ranks'], w['spec'], w['k_ms'], w['k_launches'], w['k_flops'], 10,
                               w['elapsed'], measured_traffic(dtype, True, 1.0), w['k_bytes'], executed=True)
        out['c5_dfmc'] = {'config': 'BASELINE configs[4]: Dfmc, MovieLens-style 6-relation graph (100k users x 40k movies, ratings '
                                    '98% masked and kept as known-entry lists, five 0/1 relations, two constraints)',
                          'value': 10 / w['elapsed'], 'unit': 'iters/s', 'steps': 10, 'warmup': 3,
                          'ms_per_step': w['elapsed'] / 10 * 1e3, 'dtype': dtype, 'rmse_completed': w['rmse'],
                          'launches_per_step': w['launches_per_step'], 'roofline': compact_roofline(roof),
                          '_w': {k: w[k] for k in ('n', 'ranks', 'spec', 'k_ms', 'k_launches', 'k_flops', 'elapsed', 'k_bytes')}}
    except Exception as exc:
        out['c5_dfmc'] = {'error': str(exc)[:300]}
    try:
        w = run_workload('c3', dtype, 30, 0, data='planted')
        floor = 0.01 / np.sqrt(12.0)
        out['c3_planted'] = {'config': 'BASELINE configs[2] on planted data: R = G* S* G*^T / mean + 0.01 U', 'iters': 30,
                             'dtype': dtype, 'rmse': w['rmse'], 'noise_floor': floor,
                             'rmse_over_floor': {k: v / floor for k, v in w['rmse'].items()}}
        if w.get('quantisation'):
            # bf16 storage of R ~ 1 adds q = ||bf16(R) - R||_F / sqrt(cells) of noise the fp64 reference does not see; the
            # reference itself, fed the rounded relations, lands on RMSE^2 = RMSE_f64^2 + q^2 (tests/golden/c3_planted_scaled.npz)
            q = w['quantisation']
            out['c3_planted']['quantisation'] = q
            out['c3_planted']['rmse_without_quantisation_over_floor'] = {
                k: float(np.sqrt(max(v * v - q[k] * q[k], 0.0))) / floor for k, v in w['rmse'].items()}
            out['c3_planted']['reference_at_1_25_scale_over_floor'] = {'fp64 relations': [1.1405, 1.2973, 1.1344],
                                                                        'bf16-rounded relations': [1.3005, 1.4392, 1.2963]}
    except Exception as exc:
        out['c3_planted'] = {'error': str(exc)[:300]}
    out['rank_of_8'] = rank_of_8_record(dtype)
    out['c3_tenth'] = mid_size_record(dtype)
    return out


def dicty_record():
    try:
        return dict(bench_dicty(), config='BASELINE configs[1]')
    except Exception as exc:
        return {'error': str(exc)[:300]}


def other_workloads_host(out, dtype='bf16', cpu=True, pmc=True):
    """The host legs of `workloads` (see other_workloads): the config-5 oracle at 1/2 linear scale, measured; then config 5's
    HBM traffic from PMC children of THIS run (its roofline record rebuilt with it)."""
    c5 = out.get('c5_dfmc') or {}
    w = c5.pop('_w', None)
    if 'error' in c5 or w is None:
        return out
    if cpu:
        try:
            c5['cpu_baseline'] = cpu_baseline_c5_half() or cpu_baseline_c5()
        except Exception as exc:
            c5['cpu_baseline'] = {'error': str(exc)[:200]}
    traffic = pmc_traffic_in_run(dtype, 'c5') if pmc else None
    if traffic:
        c5['roofline'] = compact_roofline(roofline_record(dtype, w['n'], w['ranks'], w['spec'], w['k_ms'], w['k_launches'],
                                                          w['k_flops'], 10, w['elapsed'], traffic, w['k_bytes'], executed=True))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--dtype', default='bf16', choices=['f32', 'f64', 'bf16'])
    ap.add_argument('--scale', type=float, default=1.0, help='linear scale of the object counts')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--data', default='uniform', choices=['uniform', 'planted'],
                    help='c3 only. uniform (default): iid U[0,1) entries as in the reference README (RMSE floor '
                         'sqrt(1/12) at any modest rank); planted: R_ij = G*_i S*_ij G*_j^T / mean + 0.01 U, on which '
                         'the RMSE discriminates (SURVEY.md 8d)')
    ap.add_argument('--workload', default='c3', choices=['c3', 'c5'],
                    help='c3 (default): BASELINE configs[2], the metric\'s workload; c5: BASELINE configs[4], '
                         'Dfmc on the MovieLens-style 6-relation graph with masks and constraints')
    ap.add_argument('--emulate-rank', default=None, metavar='k/W',
                    help='one GPU: time the compute of rank k (or `all`: every rank in turn) of W of the ownership-sharded '
                         'fit, exchanges skipped')
    ap.add_argument('--mode', default='restarts', choices=['restarts', 'relations', 'rows', 'owned'],
                    help='N>1: one independent restart per GPU (weak scaling, no collective; default); or ONE fit '
                         '(strong scaling) with whole relations partitioned over the GPUs and an RCCL all-reduce of '
                         'the E/D accumulators per iteration (relations), or with balanced row blocks of the '
                         'relations and all-reduces of W, Q and E/D (rows), or with the rows of every object type and '
                         'the matching rows of its relations -- reduce-scatter of the partial Q, all-gather of the updated '
                         'rows (owned)')
    ap.add_argument('--cpu-full-child', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--cpu-c5-child', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--parity-out', default=None, help=argparse.SUPPRESS)
    ap.add_argument('--cpu-baseline', default='auto', choices=['auto', 'full', 'sample'],
                    help='cpu_baseline leg: oracle iterations at FULL size when the host can hold them (auto), always, or the 1/10-scale sample')
    ap.add_argument('--no-engines', action='store_true', help='skip the short f32 / f64 runs of the default record')
    ap.add_argument('--no-workloads', action='store_true', help='skip the config 5 / dicty / planted legs of the default record')
    ap.add_argument('--no-pmc', action='store_true', help='do not spawn the rocprofv3 counter passes that fill roofline.traffic '
                                                          '(the committed pass of profiles/pmc_traffic.json stands in)')
    ap.add_argument('--no-strong', action='store_true', help='N > 1, mode restarts: skip the `strong` sub-record (one fit sharded '
                                                             'by ownership over the same process group, configs 3 and 5)')
    ap.add_argument('--sustained-steps', type=int, default=1000,
                    help='default run: a second timed region of this many steps (>= 10 s), reported as `sustained`; 0 = off')
    args = ap.parse_args()
    if args.cpu_full_child:
        _full_size_child(args.parity_out)
        return
    if args.cpu_c5_child:
        _c5_full_child(args.scale)
        return

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # self-launch: one rank per GPU under torch.distributed.run (RCCL over xGMI; 127.0.0.1 rendezvous)
        import subprocess
        port = os.environ.get('MASTER_PORT', str(29500 + os.getpid() % 2000))
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', port, os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import torch
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU, or let --gpus N '
                         'launch itself)' % (args.gpus, world))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    local = local % max(torch.cuda.device_count(), 1)     # (ranks may share a GPU in smoke runs)
    torch.cuda.set_device(local)
    dist = None
    backend = os.environ.get('SKF_BENCH_BACKEND', 'nccl')   # 'gloo' only for single-GPU smoke runs
    if world > 1:
        import torch.distributed as dist
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if dist is not None:
        dist.barrier()

    c5 = (args.workload == 'c5')
    if args.emulate_rank:
        print(json.dumps(emulated_ranks(args)))
        return
    default_run = world == 1 and not c5 and args.scale == 1.0 and args.data == 'uniform'
    want_parity = default_run and not args.no_cpu_baseline
    parity_kept = {}
    # Order of a default run: the headline (quiet host) -> the full-size oracle child starts in the background -> the GPU-only
    # sub-records (engines, workloads) beside it -> the child is collected -> the other host timings -> the PMC children.
    w = run_workload(args.workload, args.dtype, args.steps, args.warmup, args.scale, args.data, args.mode, rank, world,
                     dist, backend, parity=want_parity, sustained=args.sustained_steps if default_run else 0)
    parity_kept[args.dtype] = w['parity']
    elapsed, rmse, n, spec, ranks_, types = w['elapsed'], w['rmse'], w['n'], w['spec'], w['ranks'], w['types']
    sharded = w['sharded']
    units = 1 if sharded else world        # fits advanced per step by the whole job

    out = None
    if rank == 0:
        how = {'restarts': 'one random restart per GPU',
               'relations': 'one fit, whole relations partitioned over the GPUs',
               'rows': 'one fit, balanced row blocks of the relations over the GPUs',
               'owned': 'one fit, every GPU owns the same share of the rows of every object type'}[args.mode]
        cpu_child = pfile = dicty = None
        if default_run and not args.no_workloads:
            dicty = dicty_record()             # (bound by the host's launch rate: measured before the oracle child loads the host)
        if world == 1 and not args.no_cpu_baseline and not c5:
            import tempfile
            pfile = os.path.join(tempfile.gettempdir(), 'skf_parity_%d.npz' % os.getpid()) if want_parity else None
            if default_run and not (args.no_engines and args.no_workloads):
                cpu_child = cpu_baseline_start({'auto': 'auto', 'full': True, 'sample': False}[args.cpu_baseline], pfile)
        pmc = mfma_pmc = None
        if default_run and not args.no_pmc and cpu_child is None:
            pmc = pmc_traffic_in_run(args.dtype)              # counters of THIS run when the box has rocprofv3
            mfma_pmc = None if c5 else pmc_mfma_in_run(args.dtype)
        roof = roofline_record(args.dtype, n, ranks_, spec, w['k_ms'], w['k_launches'], w['k_flops'], args.steps, elapsed,
                               pmc or measured_traffic(args.dtype, c5, args.scale), w['k_bytes'], executed=c5)
        out = {
            'metric': ('DFMC update iters/sec (+ RMSE), MovieLens-style 6-relation graph with masks and constraints'
                       if c5 else
                       'DFMF update iters/sec (+ reconstruction RMSE), 3-relation graph @ ranks 128/256/256'),
            'value': units * args.steps / elapsed,
            'unit': 'iters/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'strong' if sharded else 'weak',
            'vs_baseline': None,
            'dtype': args.dtype,
            'data': 'synthetic' if (c5 or args.data == 'uniform') else 'synthetic-planted (rank-structured + 1% noise; RMSE floor 0.0029)',
            'config': {'workload': ('BASELINE configs[4]: Dfmc, MovieLens-style graph %s, ranks %s, ratings 98%% masked, '
                                    'Theta_movie = [lambda*I, sparse negative similarity], %s'
                                    % (' / '.join('%s %d' % (t, n[t]) for t in types),
                                       '/'.join(str(ranks_[t]) for t in types), how) if c5 else
                                    'BASELINE configs[2]: synthetic dense 3-type graph %dx%d / %dx%d / %dx%d, '
                                    'ranks 128/256/256, Dfmf, %s'
                                    % (n['t1'], n['t2'], n['t1'], n['t3'], n['t2'], n['t3'], how)),
                       'scale': args.scale, 'restarts': units, 'mode': args.mode,
                       'alg_flops_per_iter': alg_flops(n, spec, ranks_),
                       'exchange_bytes_per_rank_and_iter': w['exchange_bytes']},
            'rmse': rmse,
            'sustained': w['sustained'],
            'roofline': roof,
            'mfma_frac': (roof.get('mfma') or {}).get('frac'),
            'hbm_frac': (roof.get('hbm_scheduled') or roof.get('hbm_algorithmic') or {}).get('frac'),
            'host': host_info(),
        }
    watchdog = None
    if out is not None:
        # from here on the headline is measured: a TERM from an impatient caller (N = 1) or from the launcher after a peer
        # died in the strong leg (N > 1) still gets the line, without the sub-records
        import signal

        def _line_and_leave(signum, frame):
            out['interrupted'] = 'signal %d during the sub-records; the headline fields are complete' % signum
            if watchdog is not None:
                watchdog.disarm()
            print(json.dumps(out, default=str), flush=True)
            os._exit(0)
        signal.signal(signal.SIGTERM, _line_and_leave)
        if default_run and not (args.no_cpu_baseline and args.no_engines and args.no_workloads and args.no_pmc):
            # ... and a process that does not survive a sub-record (a GPU fault aborts it) still leaves the headline behind
            # (not in the counter children of this file, which run no sub-record -- and run under a profiler)
            watchdog = LineWatchdog(json.dumps(dict(out, interrupted='the process died during the sub-records; the headline '
                                                                 'fields are complete'), default=str))
            if os.environ.get('SKF_BENCH_DIE_IN_SUBRECORDS'):        # rehearsal (tools/gpu_round2.sh)
                os.kill(os.getpid(), signal.SIGKILL)
    if default_run and not args.no_engines:
        # the reference computes in f64: short runs of the f32 and f64 engines on the same graph
        engines = {}
        for dt in ('f32', 'f64'):
            if dt == args.dtype:
                continue
            try:                   # (a sub-record reports its own failure: the headline line above is already measured)
                e = run_workload('c3', dt, 3, 1, parity=want_parity)
                parity_kept[dt] = e['parity']
                r = roofline_record(dt, n, ranks_, spec, e['k_ms'], e['k_launches'], e['k_flops'], 3, e['elapsed'])
                engines[dt] = {'value': 3 / e['elapsed'], 'unit': 'iters/s', 'steps': 3, 'warmup': 1,
                               'ms_per_step': e['elapsed'] / 3 * 1e3, 'rmse': e['rmse'], 'bound': r['bound'],
                               'achieved': r['achieved'], 'peak': r['peak'], 'unit_roofline': r['unit'], 'frac': r['frac']}
            except Exception as exc:
                engines[dt] = {'error': str(exc)[:300]}
        out['engines'] = engines
    if default_run and not args.no_workloads:
        out['workloads'] = other_workloads(args.dtype, dicty)
    hung = False
    if world > 1 and args.mode == 'restarts' and not args.no_strong and not c5 and args.data == 'uniform':
        # collective: every rank runs it; bounded, so that the restarts line above survives a collective that never returns,
        # and watched, so that it survives a rank 0 that does not come back at all
        if rank == 0:
            watchdog = LineWatchdog(json.dumps(dict(out, strong={'error': 'rank 0 died inside the strong leg'}), default=str))
        strong, hung = run_bounded(lambda: strong_record(args.dtype, rank, world, dist, backend, scale=args.scale),
                                   float(os.environ.get('SKF_STRONG_TIMEOUT', '300')), local)
        if rank == 0:
            watchdog.disarm()
            out['strong'] = strong
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            if c5:
                full = cpu_baseline_c5_full(args.scale) if args.cpu_baseline == 'full' else None
                out['cpu_baseline'] = full or cpu_baseline_c5(0.25 * args.scale, args.scale)
            else:
                out['cpu_baseline'] = cpu_baseline(full={'auto': 'auto', 'full': True, 'sample': False}[args.cpu_baseline],
                                                   parity_out=pfile, child=cpu_child)
                # full-size parity: the engine's first iterations against the oracle's on the same inputs (both legs ran
                # above; the oracle's only when the host could hold the fp64 graph)
                if pfile and os.path.exists(pfile) and not out['cpu_baseline'].get('projection', True):
                    out['parity_full_size'] = {dt: parity_record(pfile, kept, dt) for dt, kept in parity_kept.items() if kept}
                    out['parity_full_size']['what'] = ('engine vs NumPy oracle (reference op order, fp64) from the same counter-based R and '
                                                       'G0 at FULL size: max relative deviation of the backbones and of %d rows of every '
                                                       'factor after iterations %s, of the relation errors after iteration %d; S_gate: '
                                                       'backbone deviation / (cond(Gram_i) cond(Gram_j)) of the engine\'s own factors '
                                                       'against eps(engine) -- S = K_i W K_j amplifies a perturbation of W by the two '
                                                       'condition numbers' % (PARITY_ROWS, '/'.join(map(str, PARITY_CHECKPOINTS)), PARITY_ITERS))
                    os.remove(pfile)
                elif want_parity:
                    out['parity_full_size'] = {'error': 'no full-size oracle run on this host (cpu_baseline.projection)'}
        if default_run and 'workloads' in out:
            other_workloads_host(out['workloads'], args.dtype, cpu=not args.no_cpu_baseline, pmc=not args.no_pmc)
        if default_run and not args.no_pmc and cpu_child is not None:
            # the counters of THIS run, once the host is quiet again: two rocprofv3 children; the roofline record is rebuilt
            pmc = pmc_traffic_in_run(args.dtype)
            if pmc:
                roof = roofline_record(args.dtype, n, ranks_, spec, w['k_ms'], w['k_launches'], w['k_flops'], args.steps,
                                       elapsed, pmc, w['k_bytes'], executed=c5)
                out.update({'roofline': roof, 'mfma_frac': (roof.get('mfma') or {}).get('frac'),
                            'hbm_frac': (roof.get('hbm_scheduled') or roof.get('hbm_algorithmic') or {}).get('frac')})
            mfma_pmc = None if c5 else pmc_mfma_in_run(args.dtype)
        if mfma_pmc and isinstance(out.get('roofline'), dict):
            # matrix-core busy share of the contraction launches from the counters of THIS run, beside the hipEvent figure
            out['roofline'].update({'mfma_busy': mfma_pmc['mfma_busy'], 'clock_ghz': mfma_pmc['clock_ghz'], 'mfma_pmc': mfma_pmc})
        if PMC_ERRORS:
            out['pmc_errors'] = PMC_ERRORS[:4]
        if watchdog is not None:
            watchdog.disarm()
        try:                       # split-K launches of this process that outgrew their plan's scratch: zero, or a sizing rule is stale
            from skfusion_amd._engine import split_clamps
            out['split_clamps'] = split_clamps()
        except Exception as exc:
            out['split_clamps'] = str(exc)[:100]
        print(json.dumps(out), flush=True)
    if dist is not None:
        # N > 1: every rank leaves through os._exit once all are done -- the line is out, and no teardown (process group,
        # RCCL communicators, a strong-leg thread that may still sit in a collective) gets the chance to wait for a peer
        sys.stdout.flush()
        if not hung:
            run_bounded(dist.barrier, 60.0, local if torch.cuda.is_available() else None)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == '__main__':
    main()
