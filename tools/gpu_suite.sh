# the whole GPU suite (run on the GPU box from the repo root); the log lands in gpurun_out/suite/
set -o pipefail
mkdir -p gpurun_out/suite
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/suite/pytest.log 2>&1 || { tail -40 gpurun_out/suite/pytest.log; exit 1; }
tail -3 gpurun_out/suite/pytest.log
