set -o pipefail
mkdir -p gpurun_out/r02e
timeout -k 10 900 python -u -m pytest tests/test_parity_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread -k "fused_aggregation or midsize or teddy_crop" > gpurun_out/r02e/pytest.log 2>&1 || { tail -60 gpurun_out/r02e/pytest.log; exit 1; }
tail -3 gpurun_out/r02e/pytest.log
AB_STEPS=20 timeout -k 10 600 python tools/ab_env.py ADC_AGG_FUSE=0 ADC_AGG_FUSE=1 2>&1 | tee gpurun_out/r02e/ab_fuse.txt || exit 1
