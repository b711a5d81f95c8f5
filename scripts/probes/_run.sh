set -e
python -m pytest tests/test_hip_round3.py tests/test_hip_round2.py tests/test_hip_scorer.py tests/test_hip_signals.py tests/test_hip_index_api.py -m gpu -x -q > gpurun_out/r3_t6.log 2>&1 || { tail -60 gpurun_out/r3_t6.log; exit 1; }
tail -3 gpurun_out/r3_t6.log
python3 bench.py --config c5 > gpurun_out/r3_c5_a.json 2> gpurun_out/r3_c5_a.err || { tail -20 gpurun_out/r3_c5_a.err; exit 1; }
cat gpurun_out/r3_c5_a.json; tail -3 gpurun_out/r3_c5_a.err
