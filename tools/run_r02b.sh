set -o pipefail
mkdir -p gpurun_out/r02b
timeout -k 10 600 python -u -m pytest tests/test_harness.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r02b/pytest.log 2>&1 || { tail -40 gpurun_out/r02b/pytest.log; exit 1; }
tail -3 gpurun_out/r02b/pytest.log
AB_STEPS=20 timeout -k 10 600 python tools/ab_env.py ADC_AGG_XCD=0 ADC_AGG_XCD=1 ADC_AGG_XCD=2 2>&1 | tee gpurun_out/r02b/ab_xcd.txt || exit 1
timeout -k 10 300 bash tools/kt_size.sh r02b_teddy 450 375 64 | tee gpurun_out/r02b/teddy_kt.txt
