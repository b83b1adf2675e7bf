set -o pipefail
mkdir -p gpurun_out/r02d
timeout -k 10 900 python -u -m pytest tests/test_parity_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread -k "wta or teddy or middlebury or ragged or disparity_counts or tiny" > gpurun_out/r02d/pytest.log 2>&1 || { tail -40 gpurun_out/r02d/pytest.log; exit 1; }
tail -3 gpurun_out/r02d/pytest.log
timeout -k 10 300 bash tools/kt_size.sh r02d_teddy 450 375 64 | tee gpurun_out/r02d/teddy_kt.txt
