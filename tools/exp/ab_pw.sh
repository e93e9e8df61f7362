# usage (GPU box): bash tools/exp/ab_pw.sh: the PW instantiations (csrc/conv_gemm.hip) on / off — conv parity
# tests on both, the 1x1 shapes in isolation, the headline region
( for w in 0 1; do echo "== pw=$w"; MRCNN_TUNE=pw=$w python -m pytest tests/test_gpu_conv.py tests/test_gpu_winograd.py tests/test_gpu_split_bf16.py -x -q 2>&1 | tail -2; done
for w in 0 1 0 1; do echo "== pw=$w"; for s in "1x1"; do BENCH_TUNE=pw=$w python tools/bench_conv.py "$s" 2>/dev/null | grep -v "^shape\|^sum"; done; done ) 
bash tools/exp/ab_tune.sh "pw=0" "pw=1"
