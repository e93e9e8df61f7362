# usage: bash tools/build_variant.sh <name> "<extra hipcc flags>"   -> chainer_mask_rcnn_amd/csrc/variants/lib<name>.so
# Developer A/B builds of libmrcnn_hip.so (select with MRCNN_HIP_LIB=<path>).  The only build that
# defines MRCNN_EXPERIMENT_BUILD, without which the MRCNN_DBG_* / trace / probe switches of conv_gemm.hip
# are a compile error: an experiment library never lands in the product path.
set -e
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/chainer_mask_rcnn_amd/csrc/variants
make -C $R/chainer_mask_rcnn_amd/csrc -j8 --no-print-directory OBJDIR=$R/chainer_mask_rcnn_amd/csrc/build_$1 \
     OUT=$R/chainer_mask_rcnn_amd/csrc/variants/lib$1.so EXTRA="-DMRCNN_EXPERIMENT_BUILD $2"
