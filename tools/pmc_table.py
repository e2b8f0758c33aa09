"""Per-kernel sums of the rocprofv3 --pmc passes of tools/pmc_contraction.sh / pmc_srp.sh (rocpd sqlite or csv output) -> a table.
    python tools/pmc_table.py <dir> [substring of the kernel names to list, default gemm_bf16]"""
import glob
import os
import sqlite3
import sys


def from_db(path):
    out = {}
    db = sqlite3.connect(path)
    cur = db.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    view = next((t for t in tables if t == 'counters_collection'), None)
    if view is None:
        return out
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % view)]
    kcol = 'kernel_name' if 'kernel_name' in cols else 'name'
    for k, c, v, n in cur.execute("select %s, counter_name, sum(value), count(distinct dispatch_id) from %s group by 1, 2" % (kcol, view)):
        out.setdefault(k, {})[c] = (v, n)
    return out


def main(root, only='gemm_bf16'):
    rows = {}
    for d in sorted(glob.glob(os.path.join(root, '*'))):
        if not os.path.isdir(d):
            continue
        for db in glob.glob(os.path.join(d, '**', '*.db'), recursive=True):
            for k, cs in from_db(db).items():
                rows.setdefault(k, {}).update(cs)
    for k, cs in rows.items():
        if only not in k:
            continue
        print(k[:120])
        for c in sorted(cs):
            v, n = cs[c]
            print('    %-36s %18.6g  per dispatch %14.6g  (%d dispatches)' % (c, v, v / max(n, 1), n))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else 'gemm_bf16')
