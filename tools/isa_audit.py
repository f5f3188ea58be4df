#!/usr/bin/env python3
"""Compile each .hip with -save-temps and report per kernel: VGPRs, AGPRs, scratch bytes, LDS bytes, and
counts of scratch_/v_mfma/ds_read/global_load instructions (catches register-array demotion to scratch)."""
import os, re, subprocess, sys, tempfile
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sd-video-gen_amd", "csrc")
files = sys.argv[1:] or ["gemm", "attn", "norm", "xformer", "eltwise"]
tmp = tempfile.mkdtemp()
for f in files:
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans", "-I" + CSRC, "-c", os.path.join(CSRC, f + ".hip"),
                    "-o", os.path.join(tmp, f + ".o"), "-save-temps"], cwd=tmp, check=True, stderr=subprocess.DEVNULL)
    s = open(os.path.join(tmp, f + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    for m in re.finditer(r"^(_Z\w+):.*?\.amdhsa_kernel \1(.*?)\.end_amdhsa_kernel", s, re.S | re.M):
        name, body, desc = m.group(1), m.group(0), m.group(2)
        g = lambda k: re.search(r"\.amdhsa_" + k + r" (\d+)", desc)
        short = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", name)[:44]
        vg = g("next_free_vgpr"); sc = g("private_segment_fixed_size"); lds = g("group_segment_fixed_size"); acc = g("accum_offset")
        print("%-8s %-44s vgpr(total)=%-4s accum_off=%-4s scratch=%-4s lds=%-6s scratch_ops=%-3d mfma=%-3d ds_read=%-3d gload=%d" % (
            f, short, vg.group(1) if vg else "?", acc.group(1) if acc else "-", sc.group(1) if sc else "?", lds.group(1) if lds else "?",
            len(re.findall(r"\bscratch_", body)), len(re.findall(r"v_mfma", body)), len(re.findall(r"ds_read|ds_load", body)),
            len(re.findall(r"global_load|buffer_load", body))))
