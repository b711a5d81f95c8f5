set -e
python -m pytest tests -m gpu -x -q > gpurun_out/r3_full4.log 2>&1 || { tail -40 gpurun_out/r3_full4.log; exit 1; }
tail -3 gpurun_out/r3_full4.log
python3 scripts/probes/batch_probe.py --l2 8 32 2>&1 | grep "^B=" | cut -c1-120
