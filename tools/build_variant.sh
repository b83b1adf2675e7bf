#!/bin/bash
# Build an A/B variant of the library: tools/build_variant.sh <name> <file.hip> <extra hipcc flags...>
# -> stereo-matching_amd/lib/variants/libadc_<name>.so (same objects, one file recompiled with the flags)
set -e
name=$1; file=$2; shift 2
cd "$(dirname "$0")/../stereo-matching_amd"
make -s -j8
mkdir -p build/variants lib/variants
base=$(basename "$file" .hip)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC "$@" -c "csrc/$base.hip" -o "build/variants/${base}_$name.o"
objs=""
for o in build/adc_pre.o build/adc_aggregate.o build/adc_scanline.o build/adc_wta.o build/adc_refine.o build/adc_api.o build/adc_harness.o; do
  if [ "$(basename $o .o)" = "$base" ]; then objs="$objs build/variants/${base}_$name.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "lib/variants/libadc_$name.so" $objs -lz
echo "built lib/variants/libadc_$name.so"
