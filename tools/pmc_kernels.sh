# SQ counters of the kernels VERDICT r02 names (fused FF, the K = 320 projection, attention d = 40, the halo conv for reference):
# one rocprofv3 --pmc pass per counter group (kernel-trace only), summarised per kernel by tools/pmc_kernels.py.
#   usage (GPU box, repo root): bash tools/pmc_kernels.sh <out.json> [problem set: r03 (default) | r04 = the MX-fp8 conv and the layer-walking launch | r05 = the kernels whose epilogue / schedule changed in round 5 | r06 = gemm_pp / conv_halo shapes of the DMA-placement change: run once with SVG_PP_DMA_M=0 SVG_HALO_DMA_M=0 exported (before) and once without (after)]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/pmc_kernels.json}
d=gpurun_out/pmck; rm -rf $d; mkdir -p $d
groups=("SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_VALU_TRANS SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_SCA SQ_LDS_DATA_FIFO_FULL" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_CYCLES")
set_=${2:-r03}
if [ "$set_" = r06 ]; then probs=("gemm 7168 1280 5120" "gemm 28672 640 2560" "conv 28 32 1920 640 0" "conv 28 16 2560 1280 0"); elif [ "$set_" = r05 ]; then probs=("ff 114688" "gemmres 114688 320 320" "gemmres 28672 640 640" "conv 28 64 320 320 0"); elif [ "$set_" = r04 ]; then probs=("convmx 28 64 320 320" "convmx 28 32 640 640" "conv 28 32 640 640 0" "walk 8" "walk 28"); else probs=("ff 114688" "gemmres 114688 320 320" "attn 28 4096 4096 40" "conv 28 64 320 320 0" "gemm 7168 1280 5120"); fi
for prob in "${probs[@]}"; do
  set -- $prob; name=$(echo $prob | tr " " "_")
  j=0
  for grp in "${groups[@]}"; do
    timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d/${name}_$j -o one -- python3 tools/kone.py $prob > $d/${name}_$j.log 2>&1 || echo "pass failed: $name / $grp"
    j=$((j+1))
  done
done
python3 tools/pmc_kernels.py $d $out "${probs[@]}"
