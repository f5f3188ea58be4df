#!/bin/bash
mkdir -p gpurun_out
for wl in cfg4 cfg1 cfg3; do
  extra=""; [ $wl = cfg4 ] && extra="--clips 28"
  timeout -k 10 550 python bench.py --workload $wl $extra --steps 1 --warmup 1 --no-fp8-extra > gpurun_out/r05_bench_$wl.log 2>&1
  grep "^{" gpurun_out/r05_bench_$wl.log > gpurun_out/r05_bench_line_$wl.json
  python -c "
import json
d=json.load(open('gpurun_out/r05_bench_line_$wl.json')); print('$wl', round(d['value'],2), d['unit'], d['dtype'], d['config']['workload'][:90])"
done
