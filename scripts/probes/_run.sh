set -e
python -m pytest tests/test_hip_round3.py tests/test_hip_mfma_f32.py -m gpu -x -q > gpurun_out/r3_t11.log 2>&1 || { tail -40 gpurun_out/r3_t11.log; exit 1; }
tail -2 gpurun_out/r3_t11.log
for B in 2 8 32; do python3 bench.py --batch $B --shadow 1 --steps 300 --warmup 60 --cpu-queries 50 > gpurun_out/r3_sh_b$B.json 2> gpurun_out/r3_sh_b$B.err || { tail -5 gpurun_out/r3_sh_b$B.err; }; python3 -c "
import json
d=json.load(open('gpurun_out/r3_sh_b$B.json')); print('B=$B shadow', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['mean_kernel_ms'], d.get('parity'))"; done
python3 bench.py --batch 32 --steps 300 --warmup 60 --cpu-queries 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('B=32 plain', d['value'], d['ms_per_step'], d['roofline']['mean_kernel_ms'])"
