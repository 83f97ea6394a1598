mkdir -p gpurun_out/cli
run() { name=$1; shift; timeout 600 python run.py "$@" > gpurun_out/cli/$name.log 2>&1; echo "$name rc=$?" >> gpurun_out/cli/summary.txt; tail -3 gpurun_out/cli/$name.log >> gpurun_out/cli/summary.txt; }
rm -f gpurun_out/cli/summary.txt
run conv3d experiment=conv3d trainer.max_epochs=1
run conv3d_of experiment=conv3d_optical_flow trainer.max_epochs=1
run example_simple experiment=example_simple trainer.max_epochs=1
run exp001 experiment=exp001_plumbing trainer.max_epochs=1
run exp003 experiment=exp003_perceiver trainer.max_epochs=1
run baseline model=baseline trainer.fast_dev_run=true
run default trainer.fast_dev_run=true
cat gpurun_out/cli/summary.txt
