set -o pipefail
mkdir -p gpurun_out/r02i
AB_STEPS=10 timeout -k 10 600 python tools/ab_bench.py base uniform 2>&1 | tee gpurun_out/r02i/ab.txt
