#!/bin/bash
# The round's bench lines WITHOUT a profiler attached (run from the repo root on the GPU box, after profile_round.sh so that
# profiles/hbm_traffic.json matches these sources):   bash scripts/bench_round.sh <tag>
# One JSON line per file under gpurun_out/bench_<tag>/: the default command (config C2), the driver's form (--steps 20
# --warmup 5), C3, C4, C5, 32 / 256 queries per step over the C2 corpus, the bf16-shadow legs (1 / 8 / 32 / 256 queries per
# step: pre-selection over the shadow + exact re-scoring; the one-query leg is an extra line, not the headline), and the
# single-GPU rehearsal of the sharded loop.
set -e
TAG=${1:-r04}
OUT=gpurun_out/bench_$TAG
mkdir -p $OUT
run() { local name=$1; shift; python3 bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err || { tail -20 $OUT/bench_$name.err; exit 1; }; cut -c1-260 $OUT/bench_$name.json; }
run c2
run c2_steps20 --steps 20 --warmup 5
run c2_steps20_condition0 --steps 20 --warmup 5 --condition-ms 0
# other embedding widths at equal corpus bytes (~3 GB, one query per step): the any-width row kernels
# (--cpu-queries 64: the parity gate — 64 queries against the oracle — runs in these lines too; the CPU figure they carry is a
# 64-query sample of THAT workload, not the headline's baseline)
# (50, 129, 301, 1001: rows that are not whole 16-byte units)
for d in 128 384 1000 1280 3072 50 129 301 1001; do
  run dim$d --dim $d --docs $(( 3072000000 / (4 * d) )) --cpu-queries 64
done
run dim384_batch4 --dim 384 --docs 2000000 --batch 4 --cpu-queries 64
# 32 queries per step at widths with a partial last chunk on the depth-split pass (round 4)
run dim384_batch32 --dim 384 --docs 2000000 --batch 32 --steps 300 --warmup 60 --cpu-queries 64
run dim640_batch32 --dim 640 --docs 1200000 --batch 32 --steps 300 --warmup 60 --cpu-queries 64
run dim128_batch32 --dim 128 --docs 6000000 --batch 32 --steps 300 --warmup 60 --cpu-queries 64
run dim1280_batch32 --dim 1280 --docs 600000 --batch 32 --steps 300 --warmup 60 --cpu-queries 64
run dim2048_batch32 --dim 2048 --docs 375000 --batch 32 --steps 300 --warmup 60 --cpu-queries 64
# rows that end inside a wave's 32-column slice of the depth-split pass (dim % 4 == 0, not % 32)
run dim1000_batch32 --dim 1000 --docs 768000 --batch 32 --steps 300 --warmup 60 --cpu-queries 64
run dim300_batch32 --dim 300 --docs 2560000 --batch 32 --steps 300 --warmup 60 --cpu-queries 64
# fp32 corpus + bf16 shadow at widths outside the dim = 256 U set (round 4)
run dim384_batch256_shadow --dim 384 --docs 2000000 --batch 256 --steps 200 --warmup 40 --cpu-queries 64
run dim384_batch32_shadow --dim 384 --docs 2000000 --batch 32 --shadow 1 --steps 200 --warmup 40 --cpu-queries 64
run dim1280_batch32_shadow --dim 1280 --docs 600000 --batch 32 --shadow 1 --steps 200 --warmup 40 --cpu-queries 64
run c3 --config c3
run c4 --config c4
run c5 --config c5
run c2_batch2 --batch 2 --steps 300 --warmup 60
run c2_batch3 --batch 3 --steps 300 --warmup 60
run c2_batch4 --batch 4 --steps 300 --warmup 60
run c2_batch32 --batch 32 --steps 400 --warmup 100
run c2_batch32_shadow --batch 32 --shadow 1 --steps 400 --warmup 100
run c2_batch8_shadow --batch 8 --shadow 1 --steps 400 --warmup 100
run c2_batch1_shadow --batch 1 --shadow 1
run c2_batch256_shadow --batch 256 --steps 400 --warmup 100
run c2_batch256_shadow_k100 --batch 256 --k 100 --steps 400 --warmup 100
# other embedding widths (not BASELINE configs): the one-query search, plain and through the opt-in shadow
run dim1024 --dim 1024 --cpu-queries 0
run dim1024_batch1_shadow --dim 1024 --batch 1 --shadow 1 --cpu-queries 0
run dim1536 --dim 1536 --cpu-queries 0
run dim1536_batch1_shadow --dim 1536 --batch 1 --shadow 1 --cpu-queries 0
DEWI_BENCH_FORCE_DIST=1 python3 bench.py --docs 125000 --steps 400 --warmup 40 --cpu-queries 0 > $OUT/bench_rccl_world1_125k.json 2> $OUT/bench_rccl_world1_125k.err
cut -c1-200 $OUT/bench_rccl_world1_125k.json
