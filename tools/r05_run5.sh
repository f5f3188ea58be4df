#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py -x -q -m gpu -k "ff or fused" 2>&1 | tail -3 || exit 1
{
for i in 1 2 3; do
  echo "== lib=no stagger (FF_STAGGER=0)"; SVG_LIB=sd-video-gen_amd/csrc/build/var_nostagger/libsvg_hip.so python tools/kbench.py ff --b 28 | tail -1
  echo "== lib=stagger"; python tools/kbench.py ff --b 28 | tail -1
done
} 2>/dev/null | tee gpurun_out/r05_ff_stagger_ab.txt
