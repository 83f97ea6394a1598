# PMC passes (counters only) over gemm_bf16x3_kernel on the weight-gradient shape: bash tools/pmc_gemm.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gemmpmc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/A -o p -- python3 $R/tools/run_gemm_only.py > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/B -o p -- python3 $R/tools/run_gemm_only.py > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/C -o p -- python3 $R/tools/run_gemm_only.py > /dev/null 2>&1
cd $R; python3 tools/pmc_kernel_summary.py gemm_bf16x3_kernel 3 $O/A $O/B $O/C > $O/pmc_gemm_bf16x3_dW_64x1024.json
find $O -name "*.csv" -size +5M -delete
cat $O/pmc_gemm_bf16x3_dW_64x1024.json
