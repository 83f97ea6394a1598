# Round-6 evidence in one call (run on the GPU box: gpurun -- 'bash tools/collect_profiles_r06.sh'); everything lands under
# gpurun_out/r06/ and is copied into profiles/r06/ by hand afterwards.  Counter passes are --pmc only (no tracing beside them).
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O $R/profiles/r06; cd /tmp; export TMPDIR=/tmp
# 1. HBM bytes per launch of the train step's kernels
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $O/write.log 2>&1
cd $R; python3 tools/pmc_traffic.py $O/fetch $O/write > $O/pmc_hbm_traffic_bench_B32.json
# 2. the advection pipeline: SQ counters per kernel + HBM bytes per kernel and batch
bash tools/pmc_flow.sh gpurun_out/r06/flowpmc > $O/flowpmc.log 2>&1
cp $O/flowpmc/pmc_flow_*.json $O/
# 3. kernel trace of the train step and of the pipeline
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 $R/bench.py --no-roofline --no-cpu-baseline --no-extras > $O/kt.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/flow -o p -- python3 $R/tools/time_flow_stages.py 32 > $O/flow_pipeline_B32_stage_times.txt 2>&1
cd $R
python3 tools/step_timeline.py $O/kt > $O/step_timeline_B32.txt
cp $(find $O/kt -name "p_kernel_stats.csv" | head -1) $O/bench_conv3d_B32_kernel_stats.csv
cp $(find $O/flow -name "p_kernel_stats.csv" | head -1) $O/flow_pipeline_B32_kernel_stats.csv
# 4. the default bench line, reading the counter files just made (bench.py looks under profiles/r06/)
cp $O/pmc_hbm_traffic_bench_B32.json $O/pmc_flow_*.json profiles/r06/
python3 bench.py > $O/bench_conv3d_B32_default_run.json 2> $O/bench_default.err
python3 tools/time_fp32_step.py > $O/fp32_step_kernels.txt 2>&1
bash tools/trace_fp32_step.sh > /dev/null 2>&1; cp gpurun_out/fp32_trace/kernel_stats.txt $O/fp32_step_kernel_stats.txt
bash tools/pmc_conv.sh gpurun_out/r06/convpmc > $O/convpmc.log 2>&1; cp $O/convpmc/pmc_conv_*.json $O/
python3 tools/probes/sclk_under_kernels.py > $O/sclk_under_kernels.txt 2>&1
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*counter_collection.csv" -size +5M -delete
rm -rf $O/fetch $O/write $O/flowpmc/A $O/flowpmc/B $O/flowpmc/C $O/flowpmc/fetch $O/flowpmc/write
ls -la $O
tail -1 $O/bench_conv3d_B32_default_run.json | cut -c1-700
