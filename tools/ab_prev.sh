# (lib/variants/libadc_prev.so = the library with csrc/adc_aggregate.hip of the previous commit, built like tools/build_variant.sh does)
# parity of the current build on the fused-aggregation shapes, then timing against lib/variants/libadc_prev.so (alternating)
set -o pipefail
mkdir -p gpurun_out/ab
timeout -k 10 400 python -u -m pytest tests/test_parity_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread \
  -k "test_fused_aggregation_pairs or test_full_resolution_synthetic_bench_pair or test_aggregated_volume_midsize or test_cost_fused_first_pass or test_middlebury_2003_full" > gpurun_out/ab/pytest.log 2>&1 \
  || { tail -30 gpurun_out/ab/pytest.log; exit 1; }
tail -1 gpurun_out/ab/pytest.log
timeout -k 10 500 python tools/ab_bench.py prev base prev base 2>&1 | tee gpurun_out/ab/ab.txt
