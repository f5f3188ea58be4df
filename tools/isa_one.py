#!/usr/bin/env python3
"""Register / instruction audit of one .hip file with explicit flags: tools/isa_one.py <file.hip> [extra hipcc flags...]"""
import os, re, subprocess, sys, tempfile
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sd-video-gen_amd", "csrc")
f = sys.argv[1]; extra = sys.argv[2:]
tmp = tempfile.mkdtemp()
base = os.path.splitext(os.path.basename(f))[0]
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-honor-nans", "-I" + CSRC, "-c", os.path.join(CSRC, f), "-o", tmp + "/f.o", "-save-temps"] + extra,
               cwd=tmp, check=True, stderr=subprocess.DEVNULL)
s = open(os.path.join(tmp, base + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
for m in re.finditer(r"^(_Z\w+):(.*?)\.amdhsa_kernel \1(.*?)\.end_amdhsa_kernel", s, re.S | re.M):
    name, body, d = m.group(1), m.group(2), m.group(3)
    g = lambda k: re.search(r"\.amdhsa_" + k + r" (\d+)", d).group(1)
    print(name[-60:], "vgpr+agpr", g("next_free_vgpr"), "accum_off", g("accum_offset"), "scratch", g("private_segment_fixed_size"),
          "| mfma", len(re.findall(r"v_mfma", body)), "acc_read", len(re.findall(r"v_accvgpr_read", body)), "acc_write", len(re.findall(r"v_accvgpr_write", body)),
          "scratch_ops", len(re.findall(r"scratch_", body)), "ds_read", len(re.findall(r"ds_read", body)), "pk_fma", len(re.findall(r"v_pk_fma_f32", body)))
