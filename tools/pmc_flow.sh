# counters of the advection pipeline's kernels (tools/time_flow_stages.py 32 on the section-8(d) input): three SQ passes,
# then FETCH_SIZE and WRITE_SIZE in passes of their own.  Run on the GPU box: bash tools/pmc_flow.sh [outdir]
R=$GRAFT_REPO_ROOT; O=$R/${1:-gpurun_out/r04flowpmc}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/A -o p -- python3 $R/tools/time_flow_stages.py 32 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/B -o p -- python3 $R/tools/time_flow_stages.py 32 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/C -o p -- python3 $R/tools/time_flow_stages.py 32 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python3 $R/tools/time_flow_stages.py 32 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python3 $R/tools/time_flow_stages.py 32 > /dev/null 2>&1
cd $R
for k in "fb_prep_polyexp_mfma_kernel<false>" "fb_prep_polyexp_mfma_kernel<true>" "fb_level_u_kernel<1, false>" "fb_level_u_kernel<2, true>" prepare_stacks remap_lds weighted_mean; do python3 tools/pmc_kernel_summary.py "$k" 1 $O/A $O/B $O/C > "$O/pmc_flow_$(echo $k | tr "<>, " "____").json"; done
python3 tools/pmc_flow_traffic.py 32 $O/fetch $O/write > $O/pmc_flow_traffic_B32.json
find $O -name "*.csv" -size +5M -delete
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/pmc_flow_*.json")):
    d=json.load(open(f))
    if "counters_per_dispatch" not in d: print(json.dumps(d)[:1500]); continue
    c=d["counters_per_dispatch"]
    print(d["kernel"], d["dispatches_averaged"], d["derived"])
    for k in ("GRBM_GUI_ACTIVE","SQ_WAVE_CYCLES","SQ_BUSY_CYCLES","SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS","SQ_INSTS_MFMA","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS","SQ_LDS_BANK_CONFLICT","SQ_LDS_IDX_ACTIVE","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_WAIT_INST_LDS","SQ_ACTIVE_INST_VMEM","SQ_ACTIVE_INST_SCA","SQ_VALU_MFMA_BUSY_CYCLES"):
        if k in c: print("   ", k, round(c[k]/1e6,3), "M")
PY
