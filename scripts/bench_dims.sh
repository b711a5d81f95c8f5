#!/bin/bash
# One-query search at equal corpus bytes (~3 GB fp32) across embedding widths — run on the GPU box from the repo root:
#   bash scripts/bench_dims.sh <out-dir> [dims...]
# Per dim: one bench.py line (event-timed scan, roofline object) and a rocprofv3 kernel trace of a short run.
set -e
OUT=${1:-gpurun_out/dims}; shift || true
DIMS=${@:-"128 384 1000 1280 3072"}
mkdir -p $OUT
export TMPDIR=/tmp
BYTES=3072000000
for d in $DIMS; do
  rows=$(( BYTES / (4 * d) ))
  python3 bench.py --dim $d --docs $rows --steps 300 --warmup 30 --cpu-queries 64 --latency-queries 0 > $OUT/bench_dim$d.json 2> $OUT/bench_dim$d.err || { tail -5 $OUT/bench_dim$d.err; exit 1; }
  python3 - "$OUT/bench_dim$d.json" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rf = r["roofline"]
print(r["config"].get("workload"), "ms/step", r["ms_per_step"], "kernel", rf.get("kernel"), "scan_ms", rf.get("kernel_ms"), "frac", rf["frac"])
PY
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_dim$d -o bench -- python3 bench.py --dim $d --docs $rows --steps 100 --warmup 10 --cpu-queries 0 --latency-queries 0 > $OUT/bench_under_trace_dim$d.json 2> $OUT/trace_dim$d.err || { tail -20 $OUT/trace_dim$d.err; exit 1; }
  find $OUT/trace_dim$d -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_dim$d.csv
  head -4 $OUT/kernel_stats_dim$d.csv
  rm -rf $OUT/trace_dim$d
done
