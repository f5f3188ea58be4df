# ablation of the halo conv (PP=0 lock-step, PP=1 ping-pong): 0 full, 1 no stores, 2 no MFMA, 3 no DMA, 4 no fragment reads, 6 no barriers
# needs tools/build_abl.sh 1 2 3 4 6 first
R=$(pwd)
for d in 0 1 2 3 4 6 0; do
  if [ $d = 0 ]; then L=$R/sd-video-gen_amd/libsvg_hip.so; else L=$R/sd-video-gen_amd/csrc/build/abl/libsvg_abl$d.so; fi
  [ -f $L ] || continue
  echo "ABL=$d"; SVG_LIB=$L SVG_HALO_PP=${PP:-0} timeout -k 10 200 python tools/kbench.py conv --b ${B:-28} 2>&1 | grep -E "^b=" | head -7 | sed -n '1p;3p;6p'; done
