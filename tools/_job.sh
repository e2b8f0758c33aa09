cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1E; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/r1E/build.log 2>&1
python tools/bench_gemm_f32.py 2>&1 | grep -v amdgpu.ids | grep f64
python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' > gpurun_out/r1E/bench_bf16.json
python tools/bench_dicty.py 2>&1 | grep dicty
