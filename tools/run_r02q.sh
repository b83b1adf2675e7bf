set -o pipefail
mkdir -p gpurun_out/r02q
timeout -k 10 900 python tools/size_sweep.py 450x375x64 640x480x128 1280x720x256 1920x1080x256 2560x1440x256 3840x2160x256 3840x2160x512 2>&1 | tee gpurun_out/r02q/size_sweep.txt &&
bash tools/kt_size.sh r02q_teddy 450 375 64 | tee gpurun_out/r02q/teddy_kernels.txt &&
timeout -k 10 600 python tools/host_path_rate.py 2>&1 | tee gpurun_out/r02q/host_path.txt
