# Types the shipped command lines on the GPU box and summarises exit codes (gpurun -- 'bash tools/cli_matrix.sh').
mkdir -p gpurun_out/cli
run() { name=$1; shift; timeout 600 python run.py "$@" > gpurun_out/cli/$name.log 2>&1; echo "$name rc=$?" >> gpurun_out/cli/summary.txt; }
rm -f gpurun_out/cli/summary.txt
run default trainer.fast_dev_run=true
for e in baseline conv3d conv3d_nwp conv3d_optical_flow conv3d_sat_nwp example_simple exp001_plumbing exp003_perceiver perceiver perceiver_conv3d_sat_nwp perceiver_sat_nwp; do
  run $e experiment=$e trainer.max_epochs=1
done
cat gpurun_out/cli/summary.txt
