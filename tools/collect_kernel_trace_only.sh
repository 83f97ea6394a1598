R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03kt; mkdir -p $O; cd $R
python3 bench.py --no-extras --no-cpu-baseline > $O/bench_noextras.json 2> $O/bench.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 $R/bench.py --no-roofline --no-cpu-baseline --no-extras > $O/kt.log 2>&1
cd $R
python3 tools/step_timeline.py $O/kt > $O/step_timeline.txt
find $O -name "*kernel_trace.csv" -size +20M -delete
python3 - <<PY
import json,csv
d=json.loads(open("$O/bench_noextras.json").read().strip().splitlines()[-1])
print("bench", d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["all_conv_kernels"])
for r in csv.DictReader(open("$O/kt/p_kernel_stats.csv")):
    if "linear_bwd_dw_dx_adam" in r["Name"]: print("rocprof", r["Calls"], r["AverageNs"])
print(open("$O/kt.log").read().strip().splitlines()[-1][:200])
PY
