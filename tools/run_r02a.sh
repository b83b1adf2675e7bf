set -o pipefail
mkdir -p gpurun_out/r02a


SKIP_TESTS= tools/gpu_profile.sh r02a
