set -o pipefail
mkdir -p gpurun_out/r02m
timeout -k 10 600 python -u -m pytest tests/test_parity_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread -k "fused_aggregation or full_resolution" > gpurun_out/r02m/pytest.log 2>&1 || { tail -40 gpurun_out/r02m/pytest.log; exit 1; }
tail -2 gpurun_out/r02m/pytest.log
timeout -k 10 200 python tools/agg_stamps.py | tee gpurun_out/r02m/agg_stamps_split.txt &&
AB_STEPS=20 timeout -k 10 600 python tools/ab_bench.py base nosplit wta128 base nosplit 2>&1 | tee gpurun_out/r02m/ab2.txt
