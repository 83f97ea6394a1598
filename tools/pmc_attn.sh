# PMC passes for the bf16 attention backward (latent / cross shapes): bash tools/pmc_attn.sh [latent cross] 
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/attnpmc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for sh in ${@:-latent cross}; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/$sh/A -o p -- python3 $R/tools/run_attn_only.py $sh bwd > /dev/null 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/$sh/B -o p -- python3 $R/tools/run_attn_only.py $sh bwd > /dev/null 2>&1
  cd $R; python3 tools/pmc_kernel_summary.py attn_bwd_bf16 1 $O/$sh/A $O/$sh/B | python3 -c "
import json,sys
d=json.load(sys.stdin); c=d['counters_per_dispatch']; print('$sh', 'wave_cycles %.1fM lds_active %.1fM conflict %.1fM insts_lds %.2fM valu %.2fM wait_any %.1fM' % tuple(c[k]/1e6 for k in ['SQ_WAVE_CYCLES','SQ_LDS_IDX_ACTIVE','SQ_LDS_BANK_CONFLICT','SQ_INSTS_LDS','SQ_INSTS_VALU','SQ_WAIT_ANY'])); print(d['derived'])"; cd /tmp
done
rm -rf $O
