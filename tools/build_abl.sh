#!/bin/bash
# Builds ablated variants of the halo conv (conv_halo.hip -DHALO_ABL=n) as separate libraries
# sd-video-gen_amd/csrc/build/abl/libsvg_abl<n>.so; select one with SVG_LIB=<path> (tools/abl_halo.sh).
set -e
cd "$(dirname "$0")/../sd-video-gen_amd/csrc"
make -j8 >/dev/null
mkdir -p build/abl
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans -w"
for n in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DHALO_ABL=$n -c conv_halo.hip -o build/abl/conv_halo_$n.o &
done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/libsvg_abl$n.so build/abl/conv_halo_$n.o $(ls build/*.o | grep -v conv_halo)
done
ls -la build/abl/*.so
