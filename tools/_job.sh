cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1l; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/r1l/build.log 2>&1
python -m pytest tests -m gpu -q -k "bf16" 2>&1 | tail -3
echo "== mfma 32x32x16"; python tools/bench_gemm_bf16.py --shapes P12,Q12,Q23 --tiles 256 --splits 0 
echo "== mfma 16x16x32"; SKF_BF16_MFMA=16 python tools/bench_gemm_bf16.py --shapes P12,Q12,Q23 --tiles 256 --splits 0
