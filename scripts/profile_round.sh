#!/bin/bash
# Round profile of the bench harness on the GPU box (run from the repo root):
#   bash scripts/profile_round.sh <tag> [commit]
# 1. rocprofv3 --kernel-trace --stats of the default bench command (c2), of --config c3 and --config c5
# 2. PMC passes, ONE COUNTER SET PER RUN and never together with a trace domain: FETCH_SIZE, WRITE_SIZE (c2, c3),
#    then the SQ counters of the matrix-core kernel (c3): LDS bank conflicts, MFMA / VALU busy, wait cycles
#    and of the depth-split pass (32 queries, scripts/probes/batch_probe.py; fp32 cosine, bf16, fp32 l2): FETCH_SIZE,
#    MFMA / VALU busy, LDS
# 3. scripts/summarize_pmc.py -> pmc_summary.{txt,json} and hbm_traffic.json (bench.py's roofline.traffic)
# The program itself (python3 bench.py) follows `--` directly: no env/bash hop under the profiler.
set -e
TAG=${1:-r04}
COMMIT=${2:-unknown}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
Q="--cpu-queries 0 --latency-queries 0"
trace() {   # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$name -o bench -- python3 bench.py "$@" $Q > $OUT/bench_under_trace_$name.json 2> $OUT/trace_$name.err || { tail -20 $OUT/trace_$name.err; exit 1; }
  find $OUT/trace_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_$name.csv
  head -12 $OUT/kernel_stats_$name.csv
}
pmc() {     # dir suffix, counters (space separated in one string), bench args...
  local name=$1; local ctr=$2; shift; shift
  if rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_$name -o bench -- python3 bench.py "$@" $Q > $OUT/bench_under_pmc_$name.json 2> $OUT/pmc_$name.err; then
    echo "pmc pass $name done"
  else
    echo "pmc pass $name FAILED (counter set: $ctr)"; tail -5 $OUT/pmc_$name.err
  fi
}
probe_trace() {   # name, probe args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$name -o bench -- python3 scripts/probes/batch_probe.py "$@" > $OUT/probe_under_trace_$name.txt 2> $OUT/trace_$name.err || { tail -20 $OUT/trace_$name.err; exit 1; }
  find $OUT/trace_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_$name.csv
  head -8 $OUT/kernel_stats_$name.csv
}
probe_pmc() {     # dir suffix, counters, probe args...
  local name=$1; local ctr=$2; shift; shift
  if rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_$name -o bench -- python3 scripts/probes/batch_probe.py "$@" > $OUT/probe_under_pmc_$name.txt 2> $OUT/pmc_$name.err; then
    echo "pmc pass $name done"
  else
    echo "pmc pass $name FAILED (counter set: $ctr)"; tail -5 $OUT/pmc_$name.err
  fi
}
trace c2 --steps 300 --warmup 30
# other embedding widths at equal corpus bytes (~3 GB): the any-width row kernels (csrc/scan_any.hpp), round 4
# (301, 1001: rows that are not whole 16-byte units — the PH = true form; their traffic includes the re-read boundary units)
DIMS="128 384 1000 1280 3072 301 1001"
for d in $DIMS; do
  rows=$(( 3072000000 / (4 * d) ))
  trace dim$d --dim $d --docs $rows --steps 200 --warmup 20
  pmc FETCH_SIZE_dim$d FETCH_SIZE --dim $d --docs $rows --steps 30 --warmup 5
done
trace c3 --config c3 --steps 40 --warmup 5
trace c5 --config c5 --steps 20 --warmup 3
trace c2b256 --batch 256 --steps 60 --warmup 10
trace c2b1s --batch 1 --shadow 1 --steps 300 --warmup 30
trace c2b32s --batch 32 --shadow 1 --steps 100 --warmup 20
pmc FETCH_SIZE_c2 FETCH_SIZE --steps 40 --warmup 5
pmc WRITE_SIZE_c2 WRITE_SIZE --steps 40 --warmup 5
pmc FETCH_SIZE_c3 FETCH_SIZE --config c3 --steps 10 --warmup 2
pmc WRITE_SIZE_c3 WRITE_SIZE --config c3 --steps 10 --warmup 2
pmc SQ_LDS_c3 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" --config c3 --steps 10 --warmup 2
pmc SQ_MFMA_c3 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES" --config c3 --steps 10 --warmup 2
pmc SQ_WAIT_c3 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" --config c3 --steps 10 --warmup 2
pmc FETCH_SIZE_c5 FETCH_SIZE --config c5 --steps 10 --warmup 2
pmc FETCH_SIZE_c2b1s FETCH_SIZE --batch 1 --shadow 1 --steps 40 --warmup 5
pmc WRITE_SIZE_c5 WRITE_SIZE --config c5 --steps 10 --warmup 2
probe_trace f32b32 32
probe_trace bf16b32 --bf16 32
probe_trace f32l2b32 --l2 32
probe_pmc FETCH_SIZE_bf16b32 FETCH_SIZE --bf16 32
probe_pmc FETCH_SIZE_f32l2b32 FETCH_SIZE --l2 32
probe_pmc FETCH_SIZE_f32b32 FETCH_SIZE 32
probe_pmc SQ_MFMA_f32b32 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" 32
probe_pmc SQ_LDS_f32b32 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" 32
probe_pmc WRITE_SIZE_f32b32 WRITE_SIZE 32
probe_pmc SQ_WAIT_f32b32 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR" 32
probe_pmc WRITE_SIZE_bf16b32 WRITE_SIZE --bf16 32
python3 scripts/summarize_pmc.py $OUT --commit $COMMIT --record "1000000x768x4xB1=scan_rows_f32" "1000000x768x2xB256=mfma_scan_bf16_s16<48, false>" \
  "6000000x128x4xB1=scan_short_rows_any<0, 8, 1, 0, 1, false>" "2000000x384x4xB1=scan_rows_any<0, 2, 3, 1, 0, 1, false>" "768000x1000x4xB1=scan_rows_any<0, 4, 1, 1, 0, 1, false>" \
  "600000x1280x4xB1=scan_rows_any<0, 5, 1, 1, 0, 1, false>" "250000x3072x4xB1=scan_rows_any<0, 12, 1, 1, 0, 1, false>" \
  "2551495x301x4xB1=scan_rows_odd_contig<0, 2, 8, 0, 1>" "767232x1001x4xB1=scan_rows_any<0, 4, 2, 1, 0, 1, true>" \
  "1000000x768x4xB32=mfma_scan_f32<false, 3, false, false, false>" "1000000x768x2xB32=mfma_scan_f32<true, 3, false, false, false>" \
  "1000000x768x2xB1=scan_rows_bf16<3, 1, 0, 1, true>" \
  "rowcos_1000000x512=row_cosine_512_kernel<2>" "fit_med_7x1000000=fit_fast_kernel<false>" "fit_mad_7x1000000=fit_fast_kernel<true>" > $OUT/pmc_summary.txt
cat $OUT/pmc_summary.txt
