#!/bin/bash
# gemm_ws 16-byte epilogue A/B (same box), integer-exact GEMM tests, fp8 placement study
set -o pipefail
mkdir -p gpurun_out
BASE=sd-video-gen_amd/csrc/build/var_base/libsvg_hip.so
python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py -x -q -m gpu 2>&1 | tail -3 || exit 1
{
for i in 1 2 3; do
  echo "== lib=HEAD~ (8-byte epilogue)"; SVG_LIB=$BASE python tools/kone_time.py gemmres 114688 320 320
  echo "== lib=new (16-byte epilogue)"; python tools/kone_time.py gemmres 114688 320 320
done
} 2>/dev/null | tee gpurun_out/r05_gemm_ws_wide_ab.txt
python tools/fp8_sites.py gpurun_out/r05_fp8_sites.json 2>&1 | grep -v Warning | tee gpurun_out/r05_fp8_sites.txt
