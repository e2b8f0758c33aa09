"""Fuzz of the host fill strategies against the REFERENCE's (fusion_graph.py:464-510), in the container that holds
/root/reference only (tools/oracle_shim.py): random plain / masked inputs with NaN, +-inf, all-False masks, non-finite
values beneath the mask.  Compares data, mask and maskedness wherever the reference does not raise.
    python tools/fuzz_fill.py [cases] [seed]
"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def random_input(rs):
    n, m = rs.randint(1, 7), rs.randint(1, 7)
    x = rs.rand(n, m)
    kind = rs.randint(0, 4)                     # 0 plain, 1 masked, 2 masked with an all-False mask, 3 masked + junk beneath
    for _ in range(rs.randint(0, 4)):
        x[rs.randint(n), rs.randint(m)] = [np.nan, np.inf, -np.inf][rs.randint(3)]
    if rs.rand() < 0.2:
        x[rs.randint(n), :] = np.nan
    if rs.rand() < 0.2:
        x[:, rs.randint(m)] = np.nan
    if kind == 0:
        return x
    if kind == 2:
        return np.ma.MaskedArray(x, mask=np.zeros((n, m), dtype=bool))
    mask = rs.rand(n, m) < 0.3
    if rs.rand() < 0.2:
        mask[rs.randint(n), :] = True
    xm = np.ma.MaskedArray(x, mask=mask)
    if kind == 3 and mask.any():
        xm.data[mask] = np.where(rs.rand(int(mask.sum())) < 0.5, np.nan, np.inf)
    return xm


def same(a, b):
    da, db = np.ma.getdata(a), np.ma.getdata(b)
    ma, mb = np.ma.getmaskarray(a), np.ma.getmaskarray(b)
    if da.shape != db.shape or not np.array_equal(ma, mb):
        return False
    if np.ma.isMaskedArray(a) != np.ma.isMaskedArray(b):
        return False
    vis = ~ma                                   # the values beneath a mask that stays are compared too where both finite
    return np.array_equal(da[vis], db[vis], equal_nan=True) and np.array_equal(da[ma], db[ma], equal_nan=True)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    import oracle_shim
    oracle_shim.load_reference()
    from skfusion.fusion.base import fusion_graph as ref
    from skfusion_amd.fusion import fusion_graph as own
    rs = np.random.RandomState(seed)
    bad = {}
    ran = 0
    for case in range(cases):
        x = random_input(rs)
        for name, rf, of in (('mean', ref.fill_mean, own.fill_mean), ('row_mean', ref.fill_row, own.fill_row),
                             ('col_mean', ref.fill_col, own.fill_col),
                             ('const', lambda a: ref.fill_const(a, 0.5), lambda a: own.fill_const(a, 0.5))):
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                try:
                    want = rf(x.copy())
                except Exception:
                    continue
                got = of(x.copy())
            ran += 1
            if not same(want, got):
                bad.setdefault(name, []).append(case)
                if len(bad[name]) <= 2:
                    print('MISMATCH', name, 'case', case, 'masked' if np.ma.isMaskedArray(x) else 'plain')
                    print(' in  ', np.ma.getdata(x).tolist(), np.ma.getmaskarray(x).tolist())
                    print(' want', np.ma.getdata(want).tolist(), np.ma.getmaskarray(want).tolist())
                    print(' got ', np.ma.getdata(got).tolist(), np.ma.getmaskarray(got).tolist())
    print('fuzz_fill: %d comparisons, mismatches: %s' % (ran, {k: len(v) for k, v in bad.items()} or 'none'))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
