set -o pipefail
SKIP_TESTS= tools/gpu_profile.sh r02h
