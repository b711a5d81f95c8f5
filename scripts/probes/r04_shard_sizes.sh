# one-query step at the shard sizes of N = 1 / 2 / 4 / 8 GPUs (strong scaling of the 1 M x 768 corpus), one GPU: scan + select
for n in 1000000 500000 250000 125000; do
  python3 scripts/tune_scan.py --docs $n --dim 768 --rounds 3 --steps 200 0:0:-1 2>/dev/null | grep -v amdgpu
done
