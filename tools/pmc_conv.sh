# SQ counters of the train step's conv kernels (matrix pipe busy, LDS array active, waits): two passes over a 3-step bench run.
# Run on the GPU box: bash tools/pmc_conv.sh [outdir]   (counter passes only, no tracing beside them)
R=$GRAFT_REPO_ROOT; O=$R/${1:-gpurun_out/r06/convpmc}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --no-calibration"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/A -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/B -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --output-format csv -d $O/C -o p -- $CMD > /dev/null 2>&1
cd $R
python3 tools/pmc_kernel_summary.py "conv3d_fwd_bf16_v3_kernel<false, false, 0, 1>" 2 $O/A $O/B $O/C > $O/pmc_conv_fwd_v3.json
python3 tools/pmc_kernel_summary.py "conv3d_fwd_bf16_v3_kernel<true, false, 0, 2>" 2 $O/A $O/B $O/C > $O/pmc_conv_dgrad_v3.json
python3 tools/pmc_kernel_summary.py "conv3d_wgrad_bf16_v2_kernel<32, false, false>" 2 $O/A $O/B $O/C > $O/pmc_conv_wgrad_v2.json
python3 tools/pmc_kernel_summary.py "conv3d_first_f32in_kernel" 2 $O/A $O/B $O/C > $O/pmc_conv_first.json
find $O -name "*.csv" -size +5M -delete
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/pmc_conv_*.json")):
    d=json.load(open(f)); c=d.get("counters_per_dispatch",{})
    print(d.get("kernel"), d.get("dispatches_averaged"), d.get("derived"))
    for k in ("GRBM_GUI_ACTIVE","SQ_BUSY_CYCLES","SQ_WAVE_CYCLES","SQ_INSTS_MFMA","SQ_VALU_MFMA_BUSY_CYCLES","SQ_INSTS_LDS","SQ_LDS_IDX_ACTIVE","SQ_LDS_BANK_CONFLICT","SQ_ACTIVE_INST_LDS","SQ_WAIT_INST_LDS","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_INSTS_VALU","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_VMEM","SQ_INST_CYCLES_VMEM"):
        if k in c: print("   ", k, round(c[k]/1e6,3), "M")
PY
