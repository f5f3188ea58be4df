#!/bin/bash
mkdir -p gpurun_out
one() { # label env
  env $2 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fp8-extra 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$1: %.3f frames/s  %.1f ms/step | UNet step %.2f ms | gemm %.1f conv %.1f attention %.1f ms per frame-group' % (d['value'], d['ms_per_step'], r['unet_step']['ms_per_call'], r['by_family']['gemm']['ms'], r['by_family']['conv3x3']['ms'], r['by_family']['attention']['ms']))"
}
{
for i in 1 2; do
  one "round-4 library (61f529b)" SVG_LIB=sd-video-gen_amd/csrc/build/var_r04/libsvg_hip.so
  one "round-5 library          " X=1
done
} | tee gpurun_out/r05_whole_bench_ab_vs_r04.txt
