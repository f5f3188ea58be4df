#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
echo "== N = 2 rehearsal on one GPU (gloo, both ranks on device 0)"
SVG_DEVICE_OVERRIDE=0 SVG_DIST_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --clips 8 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r05_bench_n2_rehearsal.log 2>&1
echo "rc=$?"; grep "^{" gpurun_out/r05_bench_n2_rehearsal.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('n_gpus', d['n_gpus'], 'ranks_seen', d['ranks_seen'], 'fps %.2f' % d['value'], 'roofline' in d, 'extras' in d)"
tail -3 gpurun_out/r05_bench_n2_rehearsal.log | cut -c1-300
python -m pytest tests -x -q -m gpu > gpurun_out/r05_gpu_tests.log 2>&1
rc=$?
tail -3 gpurun_out/r05_gpu_tests.log
cp gpurun_out/parity_margins.json gpurun_out/r05_parity_margins.json 2>/dev/null
exit $rc
