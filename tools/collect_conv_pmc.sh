# PMC passes (counters only, no tracing) for the conv kernel families, B=32.  usage: collect_conv_pmc.sh [which ...]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03pmc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
B="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
C="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
WHICH=${@:-fwd dgrad wgrad first wgrad16}
for which in $WHICH; do
  for grp in A B C; do
    eval ctr=\$$grp
    timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $O/$which/$grp -o p -- python3 $R/tools/run_conv_only.py $which > $O/${which}_$grp.log 2>&1 || echo "pass $which $grp failed" >> $O/failed.txt
  done
done
cd $R
summ() { python3 tools/pmc_kernel_summary.py "$2" $3 $O/$1/A $O/$1/B $O/$1/C > $O/$4; }
for which in $WHICH; do
  case $which in
    fwd) summ fwd "conv3d_fwd_bf16_v3_kernel<false, false>" 2 pmc_conv3d_fwd_v3_L1_B32.json;;
    dgrad) summ dgrad "conv3d_fwd_bf16_v3_kernel<true, false>" 2 pmc_conv3d_dgrad_v3_L1_B32.json;;
    wgrad) summ wgrad "conv3d_wgrad_bf16_v2_kernel<32" 2 pmc_conv3d_wgrad32_L1_B32.json;;
    wgrad16) summ wgrad16 "conv3d_wgrad_bf16_v2_kernel<16" 2 pmc_conv3d_wgrad16_L0_B32.json;;
    first) summ first "conv3d_first_f32in_kernel" 2 pmc_conv3d_first_layer_f32in_B32.json;;
  esac
done
find $O -name "*.csv" -size +5M -delete
cat $O/pmc_conv3d_*.json | grep -v "^ *\"SQ_\|GRBM"; cat $O/failed.txt 2>/dev/null
