#!/bin/bash
# tools/build_variant.sh <name> "<extra hipcc flags>"  ->  sd-video-gen_amd/csrc/build/var_<name>/libsvg_hip.so
# A second build of the library with extra -D switches (compile-time ablations / experiments); run it against the
# default build in one gpurun call with SVG_LIB=<path> (same-box A/B).
set -e
name=$1; shift
cd "$(dirname "$0")/../sd-video-gen_amd/csrc"
make -j8 OBJDIR=build/var_$name OUT=build/var_$name/libsvg_hip.so CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans -w $*" 2>&1 | grep -E "error|Error" || true
ls -la build/var_$name/libsvg_hip.so
