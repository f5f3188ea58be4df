# usage: bash tools/pmc_one.sh <tag> <kone args...>   -> FETCH_SIZE / WRITE_SIZE per launch of the named problem
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p gpurun_out/pmc1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc1/${tag}_$c -o one -- python tools/kone.py "$@" > gpurun_out/pmc1/${tag}_$c.log 2>&1 || echo FAILED $c
done
python - <<PY
import csv,collections
for c in ("FETCH_SIZE","WRITE_SIZE"):
    rows=list(csv.DictReader(open("gpurun_out/pmc1/${tag}_%s/one_counter_collection.csv"%c)))
    agg=collections.defaultdict(list)
    for r in rows:
        if "igemm" in r["Kernel_Name"] or "gemm_pp" in r["Kernel_Name"] or "conv_halo" in r["Kernel_Name"] or "attn" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print("${tag}",c,k,len(v),round(sum(v)/len(v)*1024*(2 if c=="FETCH_SIZE" else 1)/1e6,1),"MB")
PY
