# the whole GPU suite, then the frame-size ladder (run on the GPU box from the repo root)
set -o pipefail
mkdir -p gpurun_out/full
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/full/pytest.log 2>&1 || { tail -40 gpurun_out/full/pytest.log; exit 1; }
tail -2 gpurun_out/full/pytest.log
timeout -k 10 900 python tools/size_sweep.py 450x375x64 640x480x128 1280x720x256 1920x1080x256 2560x1440x256 3840x2160x256 3840x2160x512 2>&1 | tee gpurun_out/full/size_sweep.txt
