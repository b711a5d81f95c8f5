set -e
python -m pytest tests/test_hip_round3.py tests/test_hip_mfma.py tests/test_hip_mfma_f32.py tests/test_hip_full_size.py -m gpu -x -q > gpurun_out/r3_t10.log 2>&1 || { tail -40 gpurun_out/r3_t10.log; exit 1; }
tail -2 gpurun_out/r3_t10.log
for K in 10 100 250; do python3 bench.py --batch 256 --k $K --steps 200 --warmup 50 --cpu-queries 50 > gpurun_out/r3_b256_k$K.json 2> gpurun_out/r3_b256_k$K.err || { tail -5 gpurun_out/r3_b256_k$K.err; }; python3 -c "
import json,sys
d=json.load(open('gpurun_out/r3_b256_k$K.json')); print('k=$K', d['value'], d['ms_per_step'], d['roofline']['mean_kernel_ms'], d.get('parity'))"; done
