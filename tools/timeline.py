"""One iteration of a rocprofv3 kernel trace (rocpd sqlite) as a timeline: start offset, duration, stream, grid.
    python tools/timeline.py <prof_results.db> [updates_per_iteration]"""
import re
import sqlite3
import sys


def main(path, per_iter):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, grid_x, grid_y, grid_z, workgroup_x, (end-start)/1e3, start, stream_id, end "
                       "from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if 'mult_update' in r[0]]
    first, last = idx[-per_iter - 1] + 1, idx[-1]
    t0 = rows[first][6]
    for r in rows[first:last + 1]:
        n = re.sub(r'\(.*', '', r[0]).replace('skf::', '').replace('void ', '')[:72]
        print('%9.1f %8.1f us  s%-2d g=(%d,%d,%d) %s' % ((r[6] - t0) / 1e3, r[5], r[7], r[1] // max(r[4], 1), r[2], r[3], n))
    print('iteration: %.1f us' % ((rows[last][8] - t0) / 1e3))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3)
