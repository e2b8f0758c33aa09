"""Per-kernel mean/sum of the PMC counters in rocprofv3 rocpd databases (one counter per pass)."""
import re
import sqlite3
import sys


def short(name):
    return re.sub(r'\(.*$', '', name).replace('skf::', '').replace('void ', '')[:70]


for path in sys.argv[1:]:
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), sum(value), avg(grid_size_x) "
                       "from counters_collection group by kernel_name, counter_name order by 5 desc").fetchall()
    print('#', path)
    print('%-70s %-12s %6s %16s %16s' % ('kernel', 'counter', 'calls', 'mean', 'sum'))
    for name, cname, n, mean, tot, _ in rows[:14]:
        print('%-70s %-12s %6d %16.6g %16.6g' % (short(name), cname, n, mean, tot))
