set -o pipefail
mkdir -p gpurun_out/r02c
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r02c/pytest.log 2>&1 || { tail -40 gpurun_out/r02c/pytest.log; exit 1; }
tail -3 gpurun_out/r02c/pytest.log
timeout -k 10 300 bash tools/kt_size.sh r02c_teddy 450 375 64 | tee gpurun_out/r02c/teddy_kt.txt
timeout -k 10 600 python tools/size_sweep.py 450x375x64 640x480x128 1280x720x256 1920x1080x256 | tee gpurun_out/r02c/sweep.txt
