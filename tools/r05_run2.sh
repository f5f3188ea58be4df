#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_configs_gpu.py -x -q -m gpu -k "fp8 or guidance_7p5" > gpurun_out/r05_fp8_loop_parity.log 2>&1
rc=$?
grep "\[parity\]\|passed\|failed\|Error\|assert" gpurun_out/r05_fp8_loop_parity.log | tail -40
[ $rc -eq 0 ] || exit $rc
python -m pytest tests/test_transformer_gpu.py -x -q -m gpu -k "walk" 2>&1 | tail -3
