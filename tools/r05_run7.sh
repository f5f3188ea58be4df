#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r05_gpu_tests.log 2>&1
rc=$?
tail -4 gpurun_out/r05_gpu_tests.log
cp gpurun_out/parity_margins.json gpurun_out/r05_parity_margins.json 2>/dev/null
exit $rc
