#!/bin/bash
# Round 6 session 1: smoke, GPU suite, default bench (with the new MFMA counter child).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/gpu_round2.sh r6s1 smoke tests_all fullbench > gpurun_out/r6s1_session.log 2>&1
grep -E "exit|passed|failed" gpurun_out/r6s1/summary.txt | cut -c1-200
grep '^{' gpurun_out/r6s1/bench_full.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('value',d['value'],'frac',r['frac'],'mfma_busy',r.get('mfma_busy'),'clock',r.get('clock_ghz'))
print('pmc_errors',d.get('pmc_errors'))
w=d.get('workloads',{})
print('c5',w.get('c5_dfmc',{}).get('value'),'tenth',w.get('c3_tenth',{}).get('bf16',{}).get('value'))
print('gates',{k:v.get('S_gate') for k,v in d.get('parity_full_size',{}).items() if isinstance(v,dict)})
"
