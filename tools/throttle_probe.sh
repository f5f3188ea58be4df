#!/bin/bash
# Is the process being CPU-throttled by its cgroup (CFS bandwidth: quota per 100 ms period)?  Prints cpu.max and the
# nr_throttled / throttled_usec counters around each bench run.  usage: tools/throttle_probe.sh out.txt
out=${1:-gpurun_out/throttle.txt}
cg=/sys/fs/cgroup
stat() { grep -E "nr_periods|nr_throttled|throttled_usec" $cg/cpu.stat | tr '\n' ' '; }
run() { label=$1; shift; "$@" 2>&1 | grep '^{' | cut -c1-220; echo "after $label: $(stat)"; }
{
echo "nproc=$(nproc) cpu.max=$(cat $cg/cpu.max 2>/dev/null)"
python3 -c 'import torch;print("torch default threads",torch.get_num_threads());import sd_video_gen_amd._lib as L;L.load();print("after library load",torch.get_num_threads())'
echo "before: $(stat)"
run "train (library caps the pool)" python3 bench.py --train --steps 200 --no-cpu-baseline
SVG_HOST_THREADS=128 run "train SVG_HOST_THREADS=128 (the old behaviour)" python3 bench.py --train --steps 200 --no-cpu-baseline
run "sample (library caps the pool)" python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
} | tee $out
