#!/bin/bash
mkdir -p gpurun_out
run() { # env streams clips
  env $1 python bench.py --streams $2 --clips $3 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-fp8-extra 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 streams $2 clips $3: %.3f frames/s  %.1f ms/step' % (d['value'], d['ms_per_step']))"
}
{
run GPU_MAX_HW_QUEUES=4 2 56
run GPU_MAX_HW_QUEUES=8 2 56
run GPU_MAX_HW_QUEUES=8 3 57
run GPU_MAX_HW_QUEUES=8 3 84
run GPU_MAX_HW_QUEUES=8 4 56
run GPU_MAX_HW_QUEUES=8 4 112
run GPU_MAX_HW_QUEUES=4 3 84
run GPU_MAX_HW_QUEUES=4 4 112
} | tee gpurun_out/r05_streams_sweep.txt
