#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_configs_gpu.py -x -q -m gpu -k "fp8 or guidance_7p5" > gpurun_out/r05_fp8_loop_parity.log 2>&1
rc=$?
tail -3 gpurun_out/r05_fp8_loop_parity.log
[ $rc -eq 0 ] || { grep -n "Error\|assert" gpurun_out/r05_fp8_loop_parity.log | head; exit $rc; }
cp gpurun_out/parity_margins.json gpurun_out/r05_fp8_margins.json
for n in 16 24 32 40 48 56; do
  python bench.py --clips $n --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-fp8-extra 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('clips %d streams 2: %.3f frames/s  %.1f ms/step' % ($n, d['value'], d['ms_per_step']))" | tee -a gpurun_out/r05_clips_sweep.txt
done
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r05_bench_fp8extra.json 2> gpurun_out/r05_bench_fp8extra.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r05_bench_fp8extra.json') if l.startswith('{')][-1])
print('fps', d['value'], 'extras', json.dumps(d.get('extras'))[:600])
print('gemm', d['roofline']['by_family']['gemm']['frac'], 'unet', d['roofline']['unet_step'])"
