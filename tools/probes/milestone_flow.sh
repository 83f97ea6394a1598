# one GPU call for the flow milestone: tests, stage times + kernel stats, counters, stamps
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04x; mkdir -p $O; cd $R
timeout 2400 python3 -m pytest tests -q -m gpu 2>&1 | tail -6 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/flow -o p -- python3 $R/tools/time_flow_stages.py 32 > $O/flow_stage_times.txt 2>&1
cd $R
find $O -name "*kernel_trace.csv" -size +20M -delete
python3 tools/diag_stamps.py flow 32 > $O/flow_level_kernel_stamps.txt 2>&1
timeout 900 bash tools/pmc_flow.sh gpurun_out/r04x/pmc > $O/pmc_flow_summary.txt 2>&1
timeout 600 python3 tools/fuzz_flow_fused.py 7 400 2>&1 | tail -4 > $O/fuzz.txt
cat $O/flow_stage_times.txt $O/fuzz.txt; tail -30 $O/flow_level_kernel_stamps.txt
