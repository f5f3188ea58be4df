# PMC passes (one counter per run where the hardware cannot pair them, as the microarch guide prescribes) for the
# dominant conv shape, then HBM traffic of every kernel of one bench step (one clip group).  Run on the GPU box from the
# repo root; results under gpurun_out/pmc/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  echo "one $grp"
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc/one_$grp -o one -- python tools/kone.py conv 28 64 320 320 0 > gpurun_out/pmc/one_$grp.log 2>&1 || echo "FAILED $grp"
done
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  echo "bench $grp"
  timeout -k 10 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc/bench_$grp -o bench -- python bench.py --steps 1 --warmup 0 --clips 28 --streams 1 --no-cpu-baseline --no-roofline > gpurun_out/pmc/bench_$grp.log 2>&1 || echo "FAILED bench $grp"
done
find gpurun_out/pmc -name "*counter_collection.csv" | head -20
