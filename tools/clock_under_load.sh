# The clock the chip holds while the sampling loop runs: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration, per kernel family, from ONE
# rocprofv3 --pmc pass over a one-group bench step (MI355X_MICROARCH.md, DVFS give-back: the quotient reads high below ~0.3 ms dispatches).
#   usage (GPU box, repo root): bash tools/clock_under_load.sh <out.txt>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/clock_under_load.txt}
d=gpurun_out/clk; rm -rf $d; mkdir -p $d
timeout -k 10 500 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $d -o bench -- python3 bench.py --steps 1 --warmup 1 --clips 28 --streams 1 --no-cpu-baseline --no-roofline --no-fp8-extra > $d/bench.log 2>&1 || { echo "FAILED"; exit 1; }
python3 - $d $out <<'PY'
import csv, glob, sys, collections
d, out = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
fam = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(f)):
    if r.get("Counter_Name") != "GRBM_GUI_ACTIVE": continue
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    if dur < 100e-6: continue                      # short dispatches read high
    n = r["Kernel_Name"]
    k = "conv_halo" if "conv_halo_kernel" in n else "attn_dma40" if "attn_dma40" in n else "gemm_pp" if "gemm_pp" in n else "ff_pair" if "ff_pair" in n else \
        "igemm" if "igemm_kernel" in n else "gemm_ws" if "gemm_ws" in n else "xattn_fused" if "xattn_fused" in n else "gn_apply" if "gn_apply" in n else "other"
    e = fam[k]; e[0] += float(r["Counter_Value"]) / 8; e[1] += dur; e[2] += 1
with open(out, "w") as o:
    o.write("# clock under load = GRBM_GUI_ACTIVE / 8 / duration, dispatches >= 100 us of one 28-clip bench step under rocprofv3 --pmc (profiled passes run ~2.5 % slower)\n")
    for k, (cyc, dur, n) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        o.write("%-12s %5d dispatches  %8.1f ms  %.3f GHz\n" % (k, n, dur * 1e3, cyc / dur / 1e9))
print(open(out).read())
PY
