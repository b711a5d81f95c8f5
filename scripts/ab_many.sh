#!/bin/bash
# Per-kernel rocprof averages of several builds of libdewi_hip.so on ONE GPU box, two alternating rounds.
# usage: bash scripts/ab_many.sh "<libA.so> <libB.so> ..." [tune_scan args...]
LIBS=$1; shift
export TMPDIR=/tmp
for round in 1 2; do
  for L in $LIBS; do
    tag=$(basename $(dirname $L))
    out=gpurun_out/abm_${tag}_$round; rm -rf $out; mkdir -p $out
    DEWI_HIP_LIB=$PWD/$L timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o b -- python3 scripts/tune_scan.py "$@" > $out/tune.log 2>&1
    [ -f $out/b_kernel_stats.csv ] || { echo "$tag: run failed"; grep -v '^[EW]2026' $out/tune.log | head -5; exit 1; }
    python3 - $out $tag $round <<'PY'
import csv, sys
out, v, rnd = sys.argv[1:]
for r in csv.DictReader(open(f"{out}/b_kernel_stats.csv")):
    if "dewi::" in r["Name"] and int(r["Calls"]) > 50:
        print(v, rnd, r["Name"].split("(")[0][-40:], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2), "us")
PY
  done
done
