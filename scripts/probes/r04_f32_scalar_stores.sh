# fp32 32-query pass: survivor records through scalar stores (lib_sst) against vector stores (lib), interleaved on one box
set -e
P=dewi-design-for-an-entropy-weighted-index-for-text-image-corpora_amd
O=gpurun_out/r04_f32_sst; mkdir -p $O
DEWI_HIP_LIB=$PWD/$P/lib_sst/libdewi_hip.so timeout -k 10 500 python -m pytest tests/test_hip_mfma_f32.py -x -q -m gpu -k "batched_vs_oracle or overflow or shard_candidates" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for r in 1 2 3; do
  for L in lib lib_sst; do
    echo "== $L round $r"
    DEWI_HIP_LIB=$PWD/$P/$L/libdewi_hip.so python3 scripts/probes/batch_probe.py 32 2>/dev/null | grep -v amdgpu | head -2 | cut -c1-140
  done
done
