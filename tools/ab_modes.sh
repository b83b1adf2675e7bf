# A/B of the ring-1 fill forms of the fused aggregation launch (run on the GPU box from the repo root):
# parity of every mode on the shapes that use full 128-float slots, then bench.py per mode, alternating.
set -o pipefail
mkdir -p gpurun_out/modes
for m in 0 1; do
  ADC_DEBUG_ENV=1 ADC_AGG_MODE=$m timeout -k 10 400 python -u -m pytest tests/test_parity_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread \
    -k "test_fused_aggregation_pairs or test_full_resolution_synthetic_bench_pair or test_aggregated_volume_midsize" > gpurun_out/modes/pytest_mode$m.log 2>&1 \
    || { echo "mode $m FAILED"; tail -30 gpurun_out/modes/pytest_mode$m.log; exit 1; }
  echo "mode $m: $(tail -1 gpurun_out/modes/pytest_mode$m.log)"
done
timeout -k 10 600 python tools/ab_env.py ADC_AGG_MODE=0 ADC_AGG_MODE=1 2>&1 | tee gpurun_out/modes/ab.txt
