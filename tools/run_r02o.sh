set -o pipefail
mkdir -p gpurun_out/r02o
timeout -k 10 900 python -u -m pytest tests/test_parity_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread -k "teddy or ragged or disparity_counts or tiny or option_variants or full_resolution or fused_aggregation or more_rows" > gpurun_out/r02o/pytest.log 2>&1 || { tail -40 gpurun_out/r02o/pytest.log; exit 1; }
tail -2 gpurun_out/r02o/pytest.log
AB_STEPS=20 timeout -k 10 600 python tools/ab_bench.py base wta128 base 2>&1 | tee gpurun_out/r02o/ab.txt
