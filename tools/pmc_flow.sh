R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03flowpmc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/A -o p -- python3 $R/tools/time_flow_stages.py 32 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/B -o p -- python3 $R/tools/time_flow_stages.py 32 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/C -o p -- python3 $R/tools/time_flow_stages.py 32 > /dev/null 2>&1
cd $R
for k in fb_prep_polyexp_tile_kernel "fb_fused_iter_q_kernel<0, false>" "fb_fused_iter_q_kernel<1, false>" "fb_fused_iter_q_kernel<0, true>"; do python3 tools/pmc_kernel_summary.py "$k" 1 $O/A $O/B $O/C > "$O/pmc_$(echo $k | tr "<>, " "____").json"; done
find $O -name "*.csv" -size +5M -delete
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/pmc_*.json")):
    d=json.load(open(f)); c=d["counters_per_dispatch"]
    print(d["kernel"], d["dispatches_averaged"])
    for k in ("GRBM_GUI_ACTIVE","SQ_WAVE_CYCLES","SQ_BUSY_CYCLES","SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS","SQ_INSTS_MFMA","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS","SQ_LDS_BANK_CONFLICT","SQ_LDS_IDX_ACTIVE","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_WAIT_INST_LDS","SQ_ACTIVE_INST_VMEM","SQ_ACTIVE_INST_SCA","SQ_VALU_MFMA_BUSY_CYCLES"):
        if k in c: print("   ", k, round(c[k]/1e6,3), "M")
PY
