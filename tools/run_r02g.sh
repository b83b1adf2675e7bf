set -o pipefail
mkdir -p gpurun_out/r02g
export AB_STEPS=20
for f in 0 1; do
  ADC_AGG_FUSE=$f timeout -k 10 600 python tools/ab_bench.py base pipe base pipe 2>&1 | sed "s/^/fuse=$f /" | tee -a gpurun_out/r02g/ab.txt || exit 1
done
