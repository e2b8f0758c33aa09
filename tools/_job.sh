cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1u; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/r1u/build.log 2>&1
python tools/bench_gemm_f32.py 2>&1 | grep -v amdgpu.ids
