#!/bin/bash
# tools/build_prev.sh <file.hip> [git rev]: lib/variants/libadc_prev.so = the current objects with one source file
# taken from a commit (default HEAD), for A/B runs of an uncommitted change (tools/ab_bench.py prev base ...)
set -e
file=$(basename "$1" .hip); rev=${2:-HEAD}
cd "$(dirname "$0")/../stereo-matching_amd"
make -s -j8
mkdir -p build/variants lib/variants
git show "$rev:stereo-matching_amd/csrc/$file.hip" > "build/variants/${file}_prev.hip"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Icsrc -I../include -c "build/variants/${file}_prev.hip" -o "build/variants/${file}_prev.o"
objs=""
for o in build/adc_pre.o build/adc_aggregate.o build/adc_scanline.o build/adc_wta.o build/adc_refine.o build/adc_api.o build/adc_harness.o; do
  if [ "$(basename $o .o)" = "$file" ]; then objs="$objs build/variants/${file}_prev.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/variants/libadc_prev.so $objs -lz
echo "built lib/variants/libadc_prev.so ($file.hip of $rev)"
