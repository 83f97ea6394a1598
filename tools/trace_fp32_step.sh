# Kernel trace of the f32 model's train step: per-kernel averages under gpurun_out/fp32_trace/
#   gpurun -- 'bash tools/trace_fp32_step.sh'
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fp32_trace; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 $R/tools/time_fp32_step.py > $O/kt.log 2>&1
cd $R
find $O -name "*kernel_trace.csv" -size +20M -delete
python3 - <<PY
import csv, glob
f = glob.glob("$O/kt/**/p_kernel_stats.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
with open("$O/kernel_stats.txt", "w") as out:
    for r in rows[:40]:
        line = "%9.1f us x %5s  %6.2f %%  %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], float(r["Percentage"]), r["Name"][:120])
        out.write(line + "\n")
print(open("$O/kernel_stats.txt").read())
print(open("$O/kt.log").read().strip().splitlines()[1:3])
PY
