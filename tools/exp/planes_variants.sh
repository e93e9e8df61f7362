# usage (GPU box): bash tools/exp/planes_variants.sh "<planes_probe filter>" <variant> [...]
R=$GRAFT_REPO_ROOT
F="$1"; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset MRCNN_HIP_LIB; else export MRCNN_HIP_LIB=$R/chainer_mask_rcnn_amd/csrc/variants/lib$v.so; fi
  echo "=== $v"
  python $R/tools/exp/planes_probe.py "$F" 2>&1 | grep -v amdgpu.ids
done
