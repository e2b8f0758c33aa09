#!/bin/bash
# Round 5, GPU session 2: the default bench line end to end (timing of the restructured run), the wide products with the
# lower-triangle launch, longer A/Bs of the schedule switches, CU slots left to the second stream in rank-of-8 runs, and the
# two-rank smoke of `bench.py --gpus 2` with its `strong` sub-record.   tools/r05_session2.sh <out-name>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-r5s2}; OUT=gpurun_out/$N; mkdir -p $OUT
export TMPDIR=/tmp
python tools/bench_wide.py > $OUT/wide_sym.txt 2>&1; grep -E "splits=  0" $OUT/wide_sym.txt | tee -a $OUT/summary.txt
SKF_GRAM_SYM=0 python tools/bench_wide.py > $OUT/wide_nosym.txt 2>&1; grep -E "Gram.*splits=  0" $OUT/wide_nosym.txt | tee -a $OUT/summary.txt
AB_ARGS="--no-workloads --no-pmc --sustained-steps 0 --steps 100" bash tools/ab_env.sh $N/ab 3 \
   "SKF_GRAM_SYM=0 SKF_EARLY_UPDATE=0" "SKF_EARLY_UPDATE=0" "SKF_GRAM_SYM=0" "SKF_NONE=1" 2>&1 | tail -12 | tee -a $OUT/summary.txt
for v in "SKF_NONE=1" "SKF_BF16_SLOTS=240" "SKF_BF16_SLOTS=224" "SKF_BF16_SLOTS=192"; do
  for wl in c3 c5; do
    env $v timeout 300 python bench.py --emulate-rank 3/8 --steps 30 --warmup 3 --workload $wl > $OUT/emu_$wl.log 2>&1
    grep '^{' $OUT/emu_$wl.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['ranks'][0]; print('[$v] $wl rank 3/8:', round(r['compute_ms_per_step'],3), 'ms,', r['launches_per_step'], 'launches')" | tee -a $OUT/summary.txt
  done
done
( time timeout 1500 python bench.py ) > $OUT/bench_full.log 2>&1
echo "bench exit $?" | tee -a $OUT/summary.txt; grep '^{' $OUT/bench_full.log > $OUT/bench_full.json; grep real $OUT/bench_full.log | tee -a $OUT/summary.txt
python - <<PY | tee -a $OUT/summary.txt
import json
d=json.load(open('$OUT/bench_full.json'))
print('value', round(d['value'],2), 'sustained', d['sustained'] and round(d['sustained']['value'],2), 'frac', d['roofline'].get('frac'), 'traffic_kind', d['roofline'].get('traffic_kind','')[:40])
print('cpu', d['cpu_baseline'].get('value'), d['cpu_baseline'].get('sample','')[:200])
p=d.get('parity_full_size',{})
for k in ('bf16','f32','f64'):
    if k in p: print('parity', k, {a:b for a,b in p[k].items() if a in ('S_relerr','G_rows_relerr','err_relerr','S_gate')}, {c:v.get('S_relerr') for c,v in p[k]['checkpoints'].items()})
w=d['workloads']
print('c5', w['c5_dfmc'].get('value'), w['c5_dfmc'].get('cpu_baseline'), w['c5_dfmc'].get('roofline',{}).get('traffic_kind','')[:60])
print('dicty', {k:(v.get('value') if isinstance(v,dict) else v) for k,v in w['c2_dicty'].items() if k in ('f32','f64','oracle')})
print('rank_of_8', {k:(v.get('compute_ms_per_step'), v.get('launches_per_step')) for k,v in w['rank_of_8'].items() if isinstance(v,dict)})
print('c3_tenth', {k:(v.get('value') if isinstance(v,dict) else v) for k,v in w['c3_tenth'].items()})
print('engines', {k:v['value'] for k,v in d['engines'].items()})
PY
( SKF_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --scale 0.2 --no-cpu-baseline --no-engines ) > $OUT/dist_restarts_strong.log 2>&1
echo "dist restarts+strong exit $?" | tee -a $OUT/summary.txt; grep '^{' $OUT/dist_restarts_strong.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('restarts', round(d['value'],2), 'it/s; strong:', json.dumps(d.get('strong'))[:900])" | tee -a $OUT/summary.txt
echo done | tee -a $OUT/summary.txt
