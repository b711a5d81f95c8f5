set -e
python -m pytest tests -m gpu -x -q > gpurun_out/r3_full5.log 2>&1 || { tail -40 gpurun_out/r3_full5.log; exit 1; }
tail -3 gpurun_out/r3_full5.log
python3 scripts/fuzz_parity.py --big --seconds 300 --seed 7 > gpurun_out/fuzz7.log 2>&1 || true
tail -3 gpurun_out/fuzz7.log
grep -c "shadow" gpurun_out/fuzz7.log || true
