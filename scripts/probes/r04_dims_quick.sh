#!/bin/bash
# One-query bench lines at ~3 GB for a list of widths (no parity leg, no trace): bash scripts/probes/r04_dims_quick.sh <out-dir> dims...
set -e
OUT=$1; shift
mkdir -p $OUT
for d in "$@"; do
  rows=$(( 3072000000 / (4 * d) ))
  timeout -k 10 300 python3 bench.py --dim $d --docs $rows --steps 300 --warmup 30 --cpu-queries 0 --latency-queries 0 > $OUT/bench_dim$d.json 2> $OUT/bench_dim$d.err || { tail -5 $OUT/bench_dim$d.err; exit 1; }
  python3 - "$OUT/bench_dim$d.json" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rf = r["roofline"]
print("d=%s" % r["config"].get("workload").split("d=")[1].split()[0], "ms/step", r["ms_per_step"], "kernel", rf.get("kernel"), "frac", rf["frac"], flush=True)
PY
done
