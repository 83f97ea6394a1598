set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04f; mkdir -p $O; cd $R
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 $R/bench.py --no-roofline --no-cpu-baseline --no-extras > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $O/write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/flow -o p -- python3 $R/tools/time_flow_stages.py 32 > $O/flow_stage_times.txt 2>&1
cd $R
python3 tools/step_timeline.py $O/kt > $O/step_timeline.txt
python3 tools/pmc_traffic.py $O/fetch $O/write > $O/pmc.json
make -C predict_pv_yield_amd/csrc -j8 diag > /dev/null 2>&1
python3 tools/diag_stamps.py flow 32 > $O/flow_level_kernel_stamps.txt 2>&1
for k in wgrad wgrad16 fwd dgrad first; do python3 tools/diag_stamps.py $k 32; done > $O/conv_stamps.txt 2>&1
python3 tools/time_conv.py > $O/time_conv.txt 2>&1
python3 tools/time_fp32_step.py > $O/fp32_step_kernels.txt 2>&1
bash tools/probes/fullframe_run.sh > $O/flow_fullframe_704x548.txt 2>&1      # Farneback on one 25-frame 704x548 super-batch (+ kernel stats in gpurun_out/ff)
bash tools/probes/exp003_kernel_table.sh > $O/exp003_step_kernel_stats.txt 2>&1   # experiments/003 train step, kernel table
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*counter_collection.csv" -size +20M -delete
ls -la $O $O/kt $O/flow
tail -1 $O/bench_default.json | cut -c1-600
