#!/bin/bash
# Round 4: rows that are not whole 16-byte units (scan_any.hpp PH = true) — parity tests, then one-query bench lines at ~3 GB.
#   bash scripts/probes/r04_odd_rows.sh <out-dir> [dims...]
set -e
OUT=${1:-gpurun_out/odd}; shift || true
DIMS=${@:-"50 129 301 387 1001 1283 3001"}
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_hip_odd_rows.py -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
BYTES=3072000000
for d in $DIMS; do
  rows=$(( BYTES / (4 * d) ))
  timeout -k 10 300 python3 bench.py --dim $d --docs $rows --steps 300 --warmup 30 --cpu-queries 0 --latency-queries 0 > $OUT/bench_dim$d.json 2> $OUT/bench_dim$d.err || { tail -5 $OUT/bench_dim$d.err; exit 1; }
  python3 - "$OUT/bench_dim$d.json" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rf = r["roofline"]
print(r["config"].get("workload"), "ms/step", r["ms_per_step"], "kernel", rf.get("kernel"), "frac", rf["frac"], flush=True)
PY
done
