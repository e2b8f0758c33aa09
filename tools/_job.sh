cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1z; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/r1z/build.log 2>&1
python -m pytest tests -m gpu -q -k "bf16" 2>&1 | tail -3
echo "== v3 pipelined fragments (default)"; python tools/bench_gemm_bf16.py --shapes P12,Q12,P13,Q13,P23,Q23 --tiles 256 --splits 0
echo "== v2 (SKF_BF16_PIPE=0)"; SKF_BF16_PIPE=0 python tools/bench_gemm_bf16.py --shapes P12,Q12,Q23 --tiles 256 --splits 0
