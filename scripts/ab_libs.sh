#!/bin/bash
# A/B of two builds of libdewi_hip.so on ONE GPU box: per-kernel rocprof averages, alternating runs.
# usage: bash scripts/ab_libs.sh <libA.so> <libB.so> [tune_scan args...]
A=$1; B=$2; shift 2
export TMPDIR=/tmp
for round in 1 2; do
  for v in A B; do
    if [ $v = A ]; then L=$A; else L=$B; fi
    out=gpurun_out/ab_${v}_$round; rm -rf $out; mkdir -p $out
    DEWI_HIP_LIB=$PWD/$L rocprofv3 --kernel-trace --stats --output-format csv -d $out -o b -- python3 scripts/tune_scan.py "$@" > $out/tune.log 2>&1
    python3 - $out $v $round <<'PY'
import csv, sys
out, v, rnd = sys.argv[1:]
for r in csv.DictReader(open(f"{out}/b_kernel_stats.csv")):
    if "dewi::" in r["Name"] and int(r["Calls"]) > 50:
        print(v, rnd, r["Name"].split("(")[0][-40:], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2), "us")
PY
    tail -1 $out/tune.log | cut -c1-140
  done
done
