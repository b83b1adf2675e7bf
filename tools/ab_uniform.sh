# (build the variant first: tools/build_variant.sh uniform adc_aggregate.hip -DAGG_DIAG_UNIFORM)
set -o pipefail
mkdir -p gpurun_out/modes
for m in 0 1; do for v in base uniform; do echo "mode $m $v"; ADC_DEBUG_ENV=1 ADC_AGG_MODE=$m timeout -k 10 200 python tools/ab_bench.py $v || exit 1; done; done 2>&1 | tee gpurun_out/modes/uniform.txt
