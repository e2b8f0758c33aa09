#!/bin/bash
# Text summaries for profiles/ from the outputs of a GPU session (tools/gpu_round2.sh <name> smoke tests_all dist_smoke
# fullbench prof_bf16 c5_bf16 c5prof_bf16):   tools/refresh_profiles.sh <name> [round-prefix]
cd "$(dirname "$0")/.."
G=gpurun_out/$1; R=${2:-r02}
# tools/gpu_evidence.sh summarises the kernel traces on the box (the raw profiler output does not travel back); a session of
# tools/gpu_round2.sh alone leaves the databases, summarised here
for w in bf16 c5_bf16; do
  if [ -f $G/${w}_kernel_stats.txt ]; then cp $G/${w}_kernel_stats.txt profiles/${R}_${w}_kernel_stats.txt; fi
done
if [ ! -f $G/bf16_kernel_stats.txt ] && [ -f $G/prof_bf16/prof_results.db ]; then
{ echo "# rocprofv3 --kernel-trace, bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-engines (config 3 full size, the bf16 leg of the default bench.py run: 7 iterations + set-up), final build of the round"
  python tools/rocpd_summary.py $G/prof_bf16/prof_results.db 40; echo
  echo "# one iteration as a timeline (tools/timeline.py): start offset, duration, stream (s1 = main, s2 = second stream), grid"
  python tools/timeline.py $G/prof_bf16/prof_results.db 3; } > profiles/${R}_bf16_kernel_stats.txt
{ echo "# rocprofv3 --kernel-trace, bench.py --workload c5 --dtype bf16 --steps 3 --warmup 1 (config 5 full size, 4 iterations + set-up incl. torch data generation), final build of the round"
  python tools/rocpd_summary.py $G/c5prof_bf16/prof_results.db 45; echo
  echo "# one iteration as a timeline (tools/timeline.py)"
  python tools/timeline.py $G/c5prof_bf16/prof_results.db 6; } > profiles/${R}_c5_bf16_kernel_stats.txt
fi
for k in 3 5; do
  [ -f gpurun_out/$1_pmc$k/traffic.txt ] && cp gpurun_out/$1_pmc$k/traffic.txt profiles/${R}_$([ $k = 5 ] && echo c5_)bf16_pmc_traffic.txt
done
[ -f $G/dicty.txt ] && [ ! -f profiles/${R}_dicty_config2.txt ] && { echo "# python tools/bench_dicty.py / tools/bench_api_small.py, same box, final build of the round"; grep -h "dicty\|README\|NumPy" $G/dicty.txt $G/api_small.txt; } > profiles/${R}_dicty_config2.txt
[ -f $G/fuzz_known.txt ] && tail -3 $G/fuzz_known.txt > profiles/${R}_fuzz_known_entries.txt
[ -f $G/fuzz_small.txt ] && tail -3 $G/fuzz_small.txt > profiles/${R}_fuzz_small_graphs.txt
[ -f $G/fuzz_pinv.txt ] && { head -8 tools/fuzz_pinv.py | tail -7 | sed 's/^/# /'; grep -v amdgpu.ids $G/fuzz_pinv.txt | cut -c1-200; } > profiles/${R}_fuzz_pinv.txt
[ -f $G/fuzz_owned.txt ] && [ ! -f profiles/${R}_fuzz_owned_rows.txt ] && cut -c1-400 $G/fuzz_owned.txt > profiles/${R}_fuzz_owned_rows.txt
grep '^{' $G/bench_full.log > profiles/${R}_bf16_bench.json
grep '^{' $G/c5_bf16.log > profiles/${R}_c5_bf16_bench.json
{ echo "# python -m pytest tests -m gpu -q --durations=10 on the MI355X box (final build of the round)"; tail -22 $G/pytest.log; } > profiles/${R}_pytest_gpu.log
cp gpurun_out/test_deviations.txt profiles/${R}_test_deviations.txt
{ echo "# bench.py --gpus 2 (self-launched through torch.distributed.run) on the one-GPU box, two ranks sharing GPU 0 over a gloo group (SKF_BENCH_BACKEND=gloo): every multi-GPU mode end to end; the throughput of two ranks on one GPU is not a scaling number"
  echo "# (round 6) with SKF_BENCH_DIE_IN_STRONG=<rank> one rank is killed INSIDE the strong leg: the measured restarts line still comes out, exactly once"
  grep -h "^dist\|^{\|^  value" $G/summary.txt | grep -A1 "^dist" | grep -v "^--" | cut -c1-1500
  grep -h -A1 "^bench killed" $G/summary.txt | cut -c1-300; } > profiles/${R}_dist_smoke.txt
python tools/pmc_mfma_summary.py profiles/${R}_bf16_bench.json $1 > profiles/${R}_bf16_contraction_pmc.txt 2>/dev/null || rm -f profiles/${R}_bf16_contraction_pmc.txt
[ -f $G/foldin_scale.txt ] && grep fold-in $G/foldin_scale.txt > profiles/${R}_foldin_scale.txt
ls -la profiles/${R}_*
