cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1D; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/r1D/build.log 2>&1
python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' > gpurun_out/r1D/bench_bf16.json
cd /tmp && rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r1D/prof -o prof -- python $GRAFT_REPO_ROOT/bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
