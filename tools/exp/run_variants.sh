# usage (GPU box): bash tools/exp/run_variants.sh "<bench_conv filter>" <variant> [<variant> ...]
#   variant "base" = the shipped library; others = chainer_mask_rcnn_amd/csrc/variants/lib<name>.so
# Per variant: conv parity tests (tests/test_gpu_conv.py), then tools/bench_conv.py on the filter.
R=$GRAFT_REPO_ROOT
F="$1"; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset MRCNN_HIP_LIB; else export MRCNN_HIP_LIB=$R/chainer_mask_rcnn_amd/csrc/variants/lib$v.so; fi
  echo "=== $v"
  if [ -z "$SKIP_TESTS" ]; then python -m pytest $R/tests/test_gpu_conv.py $R/tests/test_gpu_winograd.py -x -q -m gpu 2>&1 | tail -2; fi
  for f in $F; do python $R/tools/bench_conv.py "$f" 2>&1 | grep -v amdgpu.ids; done
done
