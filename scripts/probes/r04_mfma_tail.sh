#!/bin/bash
# Round 4: the depth-split matrix-core pass for rows that end inside a wave's slice (dim % 4 == 0 / % 8 == 0, not % 32).
set -e
OUT=${1:-gpurun_out/mfma_tail}
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_hip_mfma_f32.py tests/test_hip_mfma.py tests/test_hip_round3.py -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
for spec in "1000 768000 32" "300 2560000 32" "100 7680000 32" "1000 768000 256"; do
  set -- $spec
  timeout -k 10 300 python3 bench.py --dim $1 --docs $2 --batch $3 --steps 100 --warmup 20 --cpu-queries 0 --latency-queries 0 --shadow 0 > $OUT/bench_dim$1_batch$3.json 2> $OUT/bench_dim$1_batch$3.err || { tail -5 $OUT/bench_dim$1_batch$3.err; exit 1; }
  python3 - "$OUT/bench_dim$1_batch$3.json" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rf = r["roofline"]
print(r["config"].get("workload")[:60], "q/s", r["value"], "ms/step", r["ms_per_step"], "kernel", rf.get("kernel"), "frac", rf["frac"], flush=True)
PY
done
