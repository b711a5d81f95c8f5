set -e
P=dewi-design-for-an-entropy-weighted-index-for-text-image-corpora_amd
for d in 128 384 640 1000; do
  rows=$(( 3072000000 / (4 * d) ))
  for L in lib lib_nq4old; do
    echo "== dim $d $L"
    DEWI_HIP_LIB=$PWD/$P/$L/libdewi_hip.so python3 scripts/tune_scan.py --docs $rows --dim $d --batch 4 --rounds 3 --steps 40 0:0:-1 2>/dev/null | grep blocks
  done
done
