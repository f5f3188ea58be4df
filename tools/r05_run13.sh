#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py tests/test_sd_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -3 || exit 1
BASE=sd-video-gen_amd/csrc/build/var_base/libsvg_hip.so
{
for i in 1 2; do
  echo "== lib=base (8-byte epilogue at BN = 160)"
  SVG_LIB=$BASE python tools/kbench.py conv --b 28 2>/dev/null | grep -E "^b=28 +(64|32|16) +[0-9]+ +(320|640|1280) 0"
  SVG_LIB=$BASE python tools/kone_time.py gemmres 28672 640 640; SVG_LIB=$BASE python tools/kone_time.py gemm 28672 1280 640; SVG_LIB=$BASE python tools/kone_time.py gemm 28672 1920 640
  echo "== lib=new (permuted-row 16-byte epilogue at BN = 160)"
  python tools/kbench.py conv --b 28 2>/dev/null | grep -E "^b=28 +(64|32|16) +[0-9]+ +(320|640|1280) 0"
  python tools/kone_time.py gemmres 28672 640 640; python tools/kone_time.py gemm 28672 1280 640; python tools/kone_time.py gemm 28672 1920 640
done
} 2>/dev/null | tee gpurun_out/r05_perm160_ab.txt
