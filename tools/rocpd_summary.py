"""Per-kernel statistics (count, total, average, share) from a rocprofv3 rocpd sqlite database
(the default --kernel-trace output of ROCm 7.2) -> plain-text summary for profiles/."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*$', '', name)
    name = name.replace('skf::', '').replace('void ', '')
    return name[:100]


def main(path, limit=25):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = 'name' if 'name' in cols else 'kernel_name'
    rows = cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by %s order by 3 desc" % (namecol, namecol)).fetchall()
    total = sum(r[2] for r in rows) or 1
    print('%-100s %7s %12s %12s %12s %12s %6s' % ('kernel', 'calls', 'total_ms', 'avg_us', 'min_us', 'max_us', '%'))
    for name, n, tot, avg, mn, mx in rows[:limit]:
        print('%-100s %7d %12.3f %12.2f %12.2f %12.2f %6.2f' % (short(name), n, tot / 1e6, avg / 1e3, mn / 1e3,
                                                                   mx / 1e3, 100.0 * tot / total))
    print('total kernel time %.3f ms over %d dispatches' % (total / 1e6, sum(r[1] for r in rows)))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25)
