#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py tests/test_sd_gpu.py -x -q -m gpu 2>&1 | tail -3 || exit 1
{
for i in 1 2; do
  echo "== lib=padding DMAs (HALO_NOPAD=0)"; SVG_LIB=sd-video-gen_amd/csrc/build/var_pad/libsvg_hip.so python tools/kbench.py conv --b 28 2>/dev/null | grep -E "^b=28 +(64|32|16) "
  echo "== lib=no padding DMAs"; python tools/kbench.py conv --b 28 2>/dev/null | grep -E "^b=28 +(64|32|16) "
done
} | tee gpurun_out/r05_halo_nopad_ab.txt
