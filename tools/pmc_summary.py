#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, `--kernel-trace` only, as
MI355X_MICROARCH.md prescribes) of one bench step into HBM bytes per kernel family and per launch.

    python tools/pmc_summary.py <dir with bench_FETCH_SIZE/ and bench_WRITE_SIZE/> <out.json> [note]

bytes = FETCH_SIZE * 1024 * 2 (gfx950: the counter tallies 128-B requests of wide streaming reads at 64 B) + WRITE_SIZE * 1024.
The summary records the source hash of the library that ran (svg_version()); bench.py uses a summary only when that hash
equals its own library's.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FAMILIES = [("conv3x3", ("conv_halo_kernel", "igemm_kernel<", "conv_")), ("gemm", ("gemm_pp_kernel", "gemm_ws_kernel", "gemm_fp8_kernel", "splitk_reduce", "ff_fused", "ff_pair")),
            ("attention", ("attn_kernel", "attn_dma", "vae_attn", "xattn_fused")), ("groupnorm", ("gn_",)), ("layernorm", ("ln_stats", "ln_finish", "layernorm_kernel")),
            ("xf", ("xf_",))]


def family(name):
    if "igemm_kernel<" in name:
        # template args <BN, AMODE[, WIDE]>: AMODE 0 = dense
        args = name.split("igemm_kernel<", 1)[1].split(">", 1)[0].split(",")
        amode = args[1].strip() if len(args) > 1 else "0"
        return "gemm" if amode.startswith("0") or "A_DENSE" in amode else "conv3x3"
    for fam, pats in FAMILIES:
        if any(p in name for p in pats):
            return fam
    return "other"


def read(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no counter_collection.csv under " + d)
    rows = []
    for f in files:
        with open(f) as fh:
            rows += [r for r in csv.DictReader(fh) if r["Counter_Name"] == counter]
    return rows


def main():
    src, out = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    from sd_video_gen_amd import _lib
    fam = collections.defaultdict(lambda: {"launches": 0, "fetch_bytes": 0.0, "write_bytes": 0.0})
    kern = collections.defaultdict(lambda: {"launches": 0, "fetch_bytes": 0.0, "write_bytes": 0.0})
    for counter, key, mul in (("FETCH_SIZE", "fetch_bytes", 2048.0), ("WRITE_SIZE", "write_bytes", 1024.0)):
        for r in read(os.path.join(src, "bench_" + counter), counter):
            name = r["Kernel_Name"]
            f = family(name)
            v = float(r["Counter_Value"]) * mul
            fam[f][key] += v
            short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:80]
            kern[short][key] += v
            if counter == "FETCH_SIZE":
                fam[f]["launches"] += 1
                kern[short]["launches"] += 1
    for d in list(fam.values()) + list(kern.values()):
        d["hbm_bytes_per_launch"] = (d["fetch_bytes"] + d["write_bytes"]) / max(d["launches"], 1)
    total = sum(d["fetch_bytes"] + d["write_bytes"] for d in fam.values())
    rec = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace), tools/pmc_step.sh; "
                     "bytes = FETCH_SIZE*1024*2 (gfx950 correction) + WRITE_SIZE*1024", "note": note,
           "src_hash": _lib.source_hash(), "total_hbm_bytes": total, "families": dict(fam),
           "kernels": dict(sorted(kern.items(), key=lambda kv: -(kv[1]["fetch_bytes"] + kv[1]["write_bytes"]))[:40])}
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
    print("total %.1f GB; " % (total / 1e9) + ", ".join("%s %.1f GB / %d" % (k, (v["fetch_bytes"] + v["write_bytes"]) / 1e9, v["launches"])
                                                         for k, v in sorted(fam.items(), key=lambda kv: -(kv[1]["fetch_bytes"] + kv[1]["write_bytes"]))))


if __name__ == "__main__":
    main()
