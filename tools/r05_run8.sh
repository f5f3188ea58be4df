#!/bin/bash
mkdir -p gpurun_out
{
for i in 1 2; do
  echo "== default (no setprio)"; python tools/kbench.py attn --b 28 2>/dev/null | grep -E "^ *(4096|1024) +(4096|1024)"
  for v in 1 2 3 4; do echo "== ATTN_PRIO=$v"; SVG_LIB=sd-video-gen_amd/csrc/build/var_aprio$v/libsvg_hip.so python tools/kbench.py attn --b 28 2>/dev/null | grep -E "^ *4096 +4096"; done
done
} | tee gpurun_out/r05_attn_prio_ab.txt
