#!/bin/bash
# Sustained graphics clock of the stand-alone contraction (P12): GRBM_GUI_ACTIVE / 8 XCDs / kernel time, product build and the
# ingest-only probe build (libskf_nomfma.so), real and all-zero operands.  tools/clock_probe.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for v in prod nomfma; do
  if [ $v = prod ]; then unset SKF_LIB_PATH; else export SKF_LIB_PATH=$PWD/scikit-fusion_amd/lib/libskf_$v.so; fi
  for z in real zero; do
    d=/tmp/clk_${v}_$z; rm -rf $d
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $d -o pmc -- python $OLDPWD/tools/bench_gemm_bf16.py --shapes P12 --tiles 256 --reps 5 $([ $z = zero ] && echo --zero) > /dev/null 2>&1 )
    python - "$d" "$v $z" <<'PY'
import glob, sqlite3, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0])
cur = db.cursor()
dur = sorted(r[0] for r in cur.execute("select (end-start)/1e3 from kernels where name like '%gemm_bf16_v2%'"))
dur = dur[len(dur) // 2]
v, n = cur.execute("select sum(value), count(distinct dispatch_id) from counters_collection where counter_name='GRBM_GUI_ACTIVE' and kernel_name like '%gemm_bf16_v2%'").fetchone()
cyc = v / n / 8.0
print('%-12s median launch %7.1f us   cycles/XCD %.3e   clock %.2f GHz' % (sys.argv[2], dur, cyc, cyc / dur / 1e3))
PY
  done
done
