#!/usr/bin/env python3
"""Summarises tools/pmc_kernels.sh: per problem, the mean per-launch value of every collected SQ counter for the problem's
dominant kernel (the one with the largest summed duration), plus the derived ratios the roofline discussion uses."""
import collections, csv, glob, json, os, sys

d, out = sys.argv[1], sys.argv[2]
res = {}
names = [a.replace(" ", "_") for a in sys.argv[3:]] or ["ff_114688", "gemmres_114688_320_320", "attn_28_4096_4096_40", "conv_28_64_320_320_0", "gemm_7168_1280_5120"]
for name in names:
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(float)
    for f in glob.glob(os.path.join(d, name + "_[0-9]*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-70:]
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    ours = {k: v for k, v in dur.items() if "at::" not in k and "elementwise" not in k}
    if not ours:
        continue
    k = max(ours, key=ours.get)
    c = {n: sum(v) / len(v) for n, v in per[k].items()}
    n_launch = max(len(v) for v in per[k].values())
    e = {"kernel": k, "launches_seen": n_launch, "counters_per_launch": c}
    g = c.get
    if g("SQ_BUSY_CYCLES") and g("SQ_VALU_MFMA_BUSY_CYCLES"):
        e["mfma_busy_over_sq_busy"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_BUSY_CYCLES")
    if g("GRBM_GUI_ACTIVE") and g("SQ_VALU_MFMA_BUSY_CYCLES"):
        e["mfma_busy_per_simd_over_gpu_cycles"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / 1024.0 / (g("GRBM_GUI_ACTIVE") / 8.0)
    if g("GRBM_GUI_ACTIVE") and g("SQ_LDS_IDX_ACTIVE"):
        e["lds_active_per_cu_over_gpu_cycles"] = g("SQ_LDS_IDX_ACTIVE") / 256.0 / (g("GRBM_GUI_ACTIVE") / 8.0)
    if g("SQ_VALU_MFMA_COEXEC_CYCLES") and g("SQ_VALU_MFMA_BUSY_CYCLES"):
        e["mfma_coexec_over_mfma_busy"] = g("SQ_VALU_MFMA_COEXEC_CYCLES") / g("SQ_VALU_MFMA_BUSY_CYCLES")
    if g("SQ_INST_CYCLES_VMEM_RD") and g("SQ_INSTS_VMEM_RD"):
        e["issue_cycles_per_vmem_rd"] = g("SQ_INST_CYCLES_VMEM_RD") / g("SQ_INSTS_VMEM_RD")
    if g("SQ_INSTS_LDS") and g("SQ_INSTS_MFMA"):
        e["lds_insts_per_mfma"] = g("SQ_INSTS_LDS") / g("SQ_INSTS_MFMA")
    if g("SQ_INSTS_VALU") and g("SQ_INSTS_MFMA"):
        e["valu_insts_per_mfma"] = (g("SQ_INSTS_VALU") - g("SQ_INSTS_MFMA")) / g("SQ_INSTS_MFMA")
    if g("SQ_WAVE_CYCLES"):
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_VMEM_RD", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC",
                  "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_VMEM_TA_ADDR_FIFO_FULL", "SQ_VMEM_TA_CMD_FIFO_FULL", "SQ_LDS_CMD_FIFO_FULL",
                  "SQ_LDS_DATA_FIFO_FULL", "SQ_INST_LEVEL_VMEM", "SQ_INST_LEVEL_LDS"):
            if g(n) is not None:
                e[n.lower() + "_over_wave_cycles"] = g(n) / g("SQ_WAVE_CYCLES")
    res[name] = e
json.dump({"source": "rocprofv3 --pmc <group> --kernel-trace, one pass per group (tools/pmc_kernels.sh); values are means over the launches of one process",
           "problems": res}, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
