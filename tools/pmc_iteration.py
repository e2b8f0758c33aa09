"""FETCH_SIZE / WRITE_SIZE of the iteration kernels of a bench.py run (tools/pmc_iteration.sh) -> bytes per iteration.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide streaming read -> x2
(MI355X_MICROARCH.md); WRITE_SIZE is taken as reported."""
import glob
import os
import re
import sqlite3
import sys

SETUP = ('fill_uniform', 'to_bf16_kernel<unsigned short>', 'at::native', 'rocclr', 'pack_', 'known_entries', 'sign_flags',
         'split_to_bf16', 'mask_zero', 'csc_sort', 'known_row_', 'known_col_', 'theta_csr_fill', 'theta_row_count', 'bits_row_count',
         'bits_csr_fill', 'csr_transpose_fill', 'csr_col_count')


def short(name):
    return re.sub(r'\(.*$', '', name).replace('skf::', '').replace('void ', '')[:64]


def main(root, iters):
    per = {}
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        for db in glob.glob(os.path.join(root, c, '**', '*.db'), recursive=True):
            cur = sqlite3.connect(db).cursor()
            for name, n, tot in cur.execute("select kernel_name, count(*), sum(value) from counters_collection "
                                            "where counter_name = ? group by kernel_name", (c,)):
                per.setdefault(short(name), {})[c] = (n, tot)
    print('%-64s %7s %14s %14s   (per iteration of %d; set-up kernels and the final residual passes listed but not summed)'
          % ('kernel', 'calls', 'read GB/iter', 'write GB/iter', iters))
    tr = tw = 0.0
    for k in sorted(per, key=lambda q: -(per[q].get('FETCH_SIZE', (0, 0))[1])):
        n, f = per[k].get('FETCH_SIZE', (0, 0.0))
        _, w = per[k].get('WRITE_SIZE', (0, 0.0))
        rd, wr = 2.0 * f * 1024 / iters / 1e9, w * 1024 / iters / 1e9
        skip = any(t in k for t in SETUP) or '0, false, 2' in k
        if not skip:
            tr += rd
            tw += wr
        if rd + wr > 0.005:
            print('%-64s %7d %14.3f %14.3f%s' % (k, n, rd, wr, '   (not summed)' if skip else ''))
    alg = float(os.environ.get('PMC_ALG_GB', '22.0'))
    print('iteration kernels: %.2f GB read + %.2f GB written per iteration; algorithmic (SURVEY 8d, one read of every '
          'relation as stored) %.1f GB -> ratio %.2f' % (tr, tw, alg, (tr + tw) / alg))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]))
