#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench command, then (separately) the HBM PMC passes.
# Run on the GPU box from the repo root:  bash scripts/profile_bench.sh <tag>
set -e
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --steps 300 --warmup 30 --cpu-queries 0 --latency-queries 0 > $OUT/bench_under_trace.json 2> $OUT/trace.err || { tail -20 $OUT/trace.err; exit 1; }
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
cat $OUT/kernel_stats.csv | head -20
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python3 bench.py --steps 40 --warmup 5 --cpu-queries 0 --latency-queries 0 > $OUT/bench_under_pmc_fetch.json 2> $OUT/pmc_fetch.err || { tail -20 $OUT/pmc_fetch.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python3 bench.py --steps 40 --warmup 5 --cpu-queries 0 --latency-queries 0 > $OUT/bench_under_pmc_write.json 2> $OUT/pmc_write.err || { tail -20 $OUT/pmc_write.err; exit 1; }
python3 scripts/summarize_pmc.py $OUT > $OUT/pmc_summary.txt
cat $OUT/pmc_summary.txt
