# usage (GPU box): bash tools/exp/ab_pw.sh: the PW / K3 instantiations (csrc/conv_gemm.hip; knob pw, bit 0 / bit 1)
# — conv parity suites on the shipped setting, the 3x3 shapes in isolation, the headline region
echo "== pw=3 (shipped)"; python -m pytest tests/test_gpu_conv.py tests/test_gpu_winograd.py tests/test_gpu_split_bf16.py -x -q 2>&1 | tail -2
for w in 1 3 1 3; do echo "== pw=$w"; BENCH_TUNE=pw=$w python tools/bench_conv.py "3x3" 2>/dev/null | grep -v "^shape\|^sum"; done
bash tools/exp/ab_tune.sh "pw=1" "pw=3"; bash tools/exp/ab_tune.sh "pw=3" "pw=1"
