# FETCH_SIZE / WRITE_SIZE of kernels that move a known 1 GiB (run on the GPU box from the repo root)
set -o pipefail
root=$(pwd); out=$root/gpurun_out/pmc_cal; mkdir -p "$out"; export TMPDIR=/tmp; cd /tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/f" -- "$root/tools/microbench/bin/pmc_calibration" > "$out/f.log" 2>&1 || { tail "$out/f.log"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/w" -- "$root/tools/microbench/bin/pmc_calibration" > "$out/w.log" 2>&1 || { tail "$out/w.log"; exit 1; }
cd "$root"
{ echo "Kernels that move exactly 1 GiB (1048576 KiB) per direction they touch (tools/microbench/pmc_calibration.hip), counters in KiB:"; python tools/pmc_summary.py "$out/f"; python tools/pmc_summary.py "$out/w"; } > "$out/summary.txt"
rm -rf "$out/f" "$out/w"
cat "$out/summary.txt"
