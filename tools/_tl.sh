export TMPDIR=/tmp
R=$PWD
python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py tests/test_gpu_model_sat_nwp.py tests/test_gpu_training.py tests/test_gpu_ddp.py -x -q -m gpu 2>&1 | tail -4
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl -o p -- python3 $R/bench.py --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > $R/gpurun_out/tl_bench.log 2>&1
cd $R
python tools/step_timeline.py gpurun_out/tl > gpurun_out/tl_timeline.txt
rm -rf gpurun_out/tl
head -8 gpurun_out/tl_timeline.txt; tail -2 gpurun_out/tl_timeline.txt
python bench.py --no-roofline --no-cpu-baseline 2>/dev/null | cut -c1-200
