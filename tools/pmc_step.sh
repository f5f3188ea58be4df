# HBM traffic of every kernel of ONE bench step of one clip group (the instrumented pass of the default bench): two
# rocprofv3 passes (FETCH_SIZE and WRITE_SIZE cannot share a pass), --kernel-trace only, program directly after `--`.
# usage (GPU box, repo root): bash tools/pmc_step.sh <clips> <out.json>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
clips=${1:-28}; out=${2:-gpurun_out/pmc_summary.json}
rm -rf gpurun_out/pmc && mkdir -p gpurun_out/pmc
for grp in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc/bench_$grp -o bench -- python3 bench.py --steps 1 --warmup 0 --clips $clips --streams 1 --no-cpu-baseline --no-roofline --no-fp8-extra > gpurun_out/pmc/bench_$grp.log 2>&1 || { echo "FAILED bench $grp"; exit 1; }
done
python3 tools/pmc_summary.py gpurun_out/pmc $out "bench.py --steps 1 --warmup 0 --clips $clips --streams 1"
rm -rf gpurun_out/pmc/bench_*/*/*.db 2>/dev/null
