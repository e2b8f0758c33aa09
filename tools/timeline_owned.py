"""One iteration of an --emulate-rank rocprofv3 kernel trace (rocpd sqlite) as a timeline: start offset, duration, stream.
The iteration is cut at the operand refresh of the last type (transpose_to_bf16_kernel<unsigned short>) before a
chol_inverse_blocked_kernel / sweep_inverse_kernel.   python tools/timeline_owned.py <prof_results.db>"""
import re
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, grid_x, grid_y, grid_z, workgroup_x, (end-start)/1e3, start, stream_id, end "
                       "from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if 'chol_inverse_blocked' in r[0] or 'sweep_inverse' in r[0]]

    def begin(i):
        while i > 0 and 'transpose_to_bf16_kernel<unsigned short>' not in rows[i - 1][0]:
            i -= 1
        return i
    s, e = begin(idx[-2]), begin(idx[-1])
    t0 = rows[s][6]
    for r in rows[s:e]:
        n = re.sub(r'\(.*', '', r[0]).replace('skf::', '').replace('void ', '')[:60]
        print('%8.1f %7.1f us s%-3d g=(%d,%d,%d) %s' % ((r[6] - t0) / 1e3, r[5], r[7], r[1] // max(r[4], 1), r[2], r[3], n))
    print('iteration %.1f us' % ((rows[e][6] - t0) / 1e3))


if __name__ == '__main__':
    main(sys.argv[1])
