# Round evidence on the GPU box (repo root): rocprofv3 kernel stats of ONE clip group running alone (the configuration of
# bench.py's instrumented pass: its per-kernel averages are uncontended), then the PMC passes (HBM traffic per family), then the
# default un-profiled bench lines.  Everything lands under gpurun_out/; the summaries are copied into profiles/ by hand.
#   usage: bash tools/prof_round.sh <tag> [clips_per_group]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r02}; clips=${2:-28}
out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
echo "== kernel trace, one group of $clips clips alone"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ktrace -o bench -- python3 bench.py --steps 1 --warmup 1 --clips $clips --streams 1 --no-cpu-baseline --no-fp8-extra > $out/bench_under_rocprof.log 2>&1 || { echo "FAILED kernel trace"; exit 1; }
find $out/ktrace -name "*kernel_stats.csv" -exec cp {} $out/${tag}_kernel_stats_one_group.csv \;
grep "^{" $out/bench_under_rocprof.log > $out/${tag}_bench_line_under_rocprof.json
rm -rf $out/ktrace
echo "== PMC passes"
bash tools/pmc_step.sh $clips $out/${tag}_pmc_summary.json || exit 1
rm -rf gpurun_out/pmc
echo "== default bench"
timeout -k 10 900 python3 bench.py > $out/${tag}_bench_default.log 2>&1 || { echo "FAILED bench"; exit 1; }
grep "^{" $out/${tag}_bench_default.log > $out/${tag}_bench_line.json
echo "== bf16 line (the default is fp16 storage), training line, shape profile"
timeout -k 10 600 python3 bench.py --dtype bf16 --no-cpu-baseline > $out/${tag}_bench_bf16.log 2>&1; grep "^{" $out/${tag}_bench_bf16.log > $out/${tag}_bench_line_bf16.json
timeout -k 10 600 python3 bench.py --train --steps 200 > $out/${tag}_bench_train.log 2>&1; grep "^{" $out/${tag}_bench_train.log > $out/${tag}_bench_line_train.json
timeout -k 10 300 python3 tools/shape_profile.py --clips $clips --steps 2 --out $out/${tag}_shape_profile.json > $out/${tag}_shape_profile.log 2>&1
echo "== no-denoise and fp8 lines"
timeout -k 10 300 python3 bench.py --no-denoise --steps 20 --warmup 3 > $out/${tag}_bench_nodenoise.log 2>&1; grep "^{" $out/${tag}_bench_nodenoise.log > $out/${tag}_bench_line_nodenoise.json
timeout -k 10 600 python3 bench.py --dtype fp8 --no-cpu-baseline > $out/${tag}_bench_fp8.log 2>&1; grep "^{" $out/${tag}_bench_fp8.log > $out/${tag}_bench_line_fp8.json
ls -la $out
echo "== the layer-walking launch of the latent Transformer alone: 48 rows (8 clips x 6 tokens), kernel trace + stamps"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/wtrace -o walk -- python3 tools/xf_walk_check.py 2048 8 4 8 8 256 > $out/${tag}_walk_48rows.log 2>&1
find $out/wtrace -name "*kernel_stats.csv" -exec cp {} $out/${tag}_walk_48rows_kernel_stats.csv \;
rm -rf $out/wtrace
for B in 1 2 4 8 14 20 28; do SVG_XF_WALK_ROWS=176 timeout -k 10 200 python3 tools/xf_walk_check.py 2048 8 4 8 $B 256 2>&1 | grep "rel-L2"; done > $out/${tag}_walk_vs_per_gemm.txt
ls -la $out
