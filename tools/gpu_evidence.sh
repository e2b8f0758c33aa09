#!/bin/bash
# Consolidated evidence session of a round, one gpurun call:   gpurun -- 'bash tools/gpu_evidence.sh <name>'
# smoke, the whole GPU suite, the default bench.py run, kernel traces of config 3 and 5, the multi-GPU smoke runs, PMC traffic
# passes and the dicty / README numbers -- everything summarised to text ON THE BOX and the raw profiler output deleted
# (gpurun copies back at most 64 MiB).  tools/refresh_profiles.sh <name> <round> then files the summaries under profiles/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-evidence}; G=gpurun_out/$N
bash tools/gpu_round2.sh $N smoke tests_all fullbench die_sub prof_bf16 c5_bf16 c5prof_bf16 dist_smoke emulate foldin > gpurun_out/${N}_session.log 2>&1
{ echo "# rocprofv3 --kernel-trace, bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-engines --no-workloads (config 3 full size, the bf16 leg of the default bench.py run: 7 iterations + set-up), final build of the round"
  python tools/rocpd_summary.py $G/prof_bf16/prof_results.db 40; echo
  echo "# one iteration as a timeline (tools/timeline.py): start offset, duration, stream (s1 = main, s2 = second stream), grid"
  python tools/timeline.py $G/prof_bf16/prof_results.db 3; } > $G/bf16_kernel_stats.txt 2>&1
{ echo "# rocprofv3 --kernel-trace, bench.py --workload c5 --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline (config 5 full size, 4 iterations + set-up incl. torch data generation), final build of the round"
  python tools/rocpd_summary.py $G/c5prof_bf16/prof_results.db 45; echo
  echo "# one iteration as a timeline (tools/timeline.py)"
  python tools/timeline.py $G/c5prof_bf16/prof_results.db 6; } > $G/c5_bf16_kernel_stats.txt 2>&1
rm -rf $G/prof_bf16 $G/c5prof_bf16
bash tools/pmc_iteration.sh ${N}_pmc3 > gpurun_out/${N}_pmc3.log 2>&1
rm -rf gpurun_out/${N}_pmc3/FETCH_SIZE gpurun_out/${N}_pmc3/WRITE_SIZE
PMC_ARGS="--workload c5" bash tools/pmc_iteration.sh ${N}_pmc5 > gpurun_out/${N}_pmc5.log 2>&1
rm -rf gpurun_out/${N}_pmc5/FETCH_SIZE gpurun_out/${N}_pmc5/WRITE_SIZE
python tools/bench_dicty.py > $G/dicty.txt 2>&1
python tools/bench_api_small.py 2>&1 | grep -v amdgpu.ids > $G/api_small.txt
timeout 300 python tools/fuzz_known.py 16 3 > $G/fuzz_known.txt 2>&1
timeout 300 python tools/fuzz_small.py 20 3 > $G/fuzz_small.txt 2>&1
timeout 600 python tools/fuzz_owned.py 40 3 > $G/fuzz_owned.txt 2>&1
timeout 600 python tools/fuzz_pinv.py 150 3 > $G/fuzz_pinv.txt 2>&1
du -sh gpurun_out | tail -1
grep -E "exit|passed|failed" $G/summary.txt | cut -c1-200
tail -1 $G/fuzz_known.txt; tail -1 $G/fuzz_small.txt; tail -1 $G/fuzz_owned.txt; tail -1 $G/fuzz_pinv.txt
