# one query over a bf16 corpus of ~1.5 GB at widths outside the dim = 256 U set: the any-width kernels, ELEM = bf16
for d in 128 384 1000 1280 3072 768; do
  rows=$(( 1536000000 / (2 * d) ))
  python3 scripts/tune_scan.py --bf16 --docs $rows --dim $d --rounds 3 --steps 60 0:0:-1 2>/dev/null | grep -v amdgpu
done
