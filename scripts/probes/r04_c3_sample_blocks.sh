set -e
P=dewi-design-for-an-entropy-weighted-index-for-text-image-corpora_amd
Q="--cpu-queries 0 --latency-queries 0"
for r in 1 2; do for L in lib_sd1 lib_sd2; do
  DEWI_HIP_LIB=$PWD/$P/$L/libdewi_hip.so python3 bench.py --config c3 $Q 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', r['value'], r['ms_per_step'], r['roofline'].get('mean_kernel_ms'))"
done; done
