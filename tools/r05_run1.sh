#!/bin/bash
# round 5, first GPU call: walk hardening tests, cooperative-launch A/B, fp8 loop-level parity
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_transformer_gpu.py -x -q -m gpu > gpurun_out/r05_walk_tests.log 2>&1 || { tail -30 gpurun_out/r05_walk_tests.log; exit 1; }
tail -3 gpurun_out/r05_walk_tests.log
for coop in 1 0; do
  for B in 1 8; do
    SVG_XF_WALK_COOP=$coop python tools/xf_walk_check.py 2048 8 4 8 $B 256 2>/dev/null | grep "rel-L2" | sed "s/^/coop=$coop /" >> gpurun_out/r05_walk_coop_ab.txt
  done
done
cat gpurun_out/r05_walk_coop_ab.txt
python -m pytest tests/test_configs_gpu.py -x -q -m gpu -k "fp8 or guidance_7p5" -s > gpurun_out/r05_fp8_loop_parity.log 2>&1
rc=$?
grep "\[parity\]\|margin\|passed\|failed\|Error" gpurun_out/r05_fp8_loop_parity.log | tail -40
exit $rc
