#!/bin/bash
# Launch-shape sweep of the any-width row kernels (GPU box, repo root): variant libraries built by scripts/build_variant.sh
# with other rows-in-flight constants x workgroups per launch, per embedding width at ~3 GB.
#   bash scripts/sweep_any.sh <out-file> "<lib dirs>" "<dims>" "<cfgs>"
OUT=$1; LIBS=$2; DIMS=$3; CFGS=${4:-"256:0:-1 512:0:-1 1024:0:-1"}
P=dewi-design-for-an-entropy-weighted-index-for-text-image-corpora_amd
: > $OUT
for d in $DIMS; do
  rows=$(( 3072000000 / (4 * d) ))
  for L in $LIBS; do
    echo "== dim $d lib $L" >> $OUT
    DEWI_HIP_LIB=$PWD/$P/$L/libdewi_hip.so python3 scripts/tune_scan.py --docs $rows --dim $d --rounds 3 --steps 60 $CFGS >> $OUT 2>&1 || exit 1
  done
done
