#!/bin/bash
# Round 4: narrow odd-width rows with a wave on CONSECUTIVE rows (scan_rows_odd_contig) — parity, bench lines, FETCH_SIZE.
set -e
OUT=${1:-gpurun_out/odd_contig}
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_hip_odd_rows.py tests/test_hip_search.py -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -n 2 $OUT/tests.log
bash scripts/probes/r04_dims_quick.sh $OUT 129 131 161 255 301 387 510
for d in 129 301; do
  rows=$(( 3072000000 / (4 * d) ))
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_dim$d -o bench -- python3 bench.py --dim $d --docs $rows --steps 30 --warmup 5 --cpu-queries 0 --latency-queries 0 > $OUT/under_pmc_dim$d.json 2> $OUT/pmc_dim$d.err || { tail -5 $OUT/pmc_dim$d.err; exit 1; }
done
python3 scripts/summarize_pmc.py $OUT 2>/dev/null | grep -i "odd_contig\|scan_rows_any" | cut -c1-200
