# C3 at sample strides 16 / 32 / 64 (variant libraries from scripts/build_variant.sh), one bench line + kernel stats each
set -e
export TMPDIR=/tmp
P=dewi-design-for-an-entropy-weighted-index-for-text-image-corpora_amd
O=gpurun_out/r04_c3_stride; mkdir -p $O
Q="--cpu-queries 0 --latency-queries 0"
for v in 32:lib 16:lib_s16 64:lib_s64 32:lib 16:lib_s16 64:lib_s64; do
  s=${v%%:*}; L=${v##*:}
  extra=""; if [ $s = 64 ]; then export DEWI_STAGE_KEYS=12288; else unset DEWI_STAGE_KEYS; fi
  DEWI_HIP_LIB=$PWD/$P/$L/libdewi_hip.so python3 bench.py --config c3 $Q > $O/bench_c3_stride$s.json 2> $O/err_$s.txt
  python3 -c "
import json
r=json.loads(open('$O/bench_c3_stride$s.json').read().strip().splitlines()[-1]); print('stride $s', r['value'], r['ms_per_step'], r['roofline'].get('frac'), r['roofline'].get('mean_kernel_ms'), r.get('parity'))"
done
for v in 32:lib 16:lib_s16 64:lib_s64; do
  s=${v%%:*}; L=${v##*:}
  if [ $s = 64 ]; then export DEWI_STAGE_KEYS=12288; else unset DEWI_STAGE_KEYS; fi
  export DEWI_HIP_LIB=$PWD/$P/$L/libdewi_hip.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$s -o bench -- python3 bench.py --config c3 --steps 60 --warmup 10 $Q > $O/under_trace_$s.json 2> $O/trace_$s.err
  find $O/trace_$s -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_c3_stride$s.csv
  rm -rf $O/trace_$s
  python3 - $O/kernel_stats_c3_stride$s.csv $s <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "dewi::" in r["Name"] and int(r["Calls"]) > 20:
        print("stride", sys.argv[2], r["Name"].split("(")[0][-44:], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2), "us")
PY
done
