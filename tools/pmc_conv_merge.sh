cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
d=gpurun_out/pmck2; rm -rf $d; mkdir -p $d
j=0
for grp in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA"; do
  for m in 2 0; do
    SVG_HALO_MERGE=$m timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d/conv${m}_$j -o one -- python3 tools/kone.py conv 28 64 320 320 0 > $d/conv${m}_$j.log 2>&1 || echo fail
  done
  j=$((j+1))
done
python3 - <<PY
import csv,glob,collections
for m in (2,0):
    per=collections.defaultdict(list)
    for f in glob.glob("gpurun_out/pmck2/conv%d_*/**/*counter_collection.csv"%m, recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv_halo" in r["Kernel_Name"]: per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    c={k:sum(v)/len(v) for k,v in per.items()}
    print("merge",m, "mfma_busy/simd/gpu_cycles %.3f"%(c["SQ_VALU_MFMA_BUSY_CYCLES"]/1024/(c["GRBM_GUI_ACTIVE"]/8)), "wait_any %.3f wait_inst %.3f active %.3f lds_stall %.3f"%(c["SQ_WAIT_ANY"]/c["SQ_WAVE_CYCLES"], c["SQ_WAIT_INST_ANY"]/c["SQ_WAVE_CYCLES"], c["SQ_ACTIVE_INST_ANY"]/c["SQ_WAVE_CYCLES"], c["SQ_WAIT_INST_LDS"]/c["SQ_WAVE_CYCLES"]), "gui_active %.3g"%c["GRBM_GUI_ACTIVE"])
PY
rm -rf gpurun_out/pmck2
