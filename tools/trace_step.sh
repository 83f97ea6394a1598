# Kernel trace of the headline train step (both arms of a tools/ab_step.py switch): per-kernel averages under gpurun_out/<name>/
#   gpurun -- 'bash tools/trace_step.sh functional.USE_RELU_MASKS masks'
R=$GRAFT_REPO_ROOT; SW=${1:-functional.USE_RELU_MASKS}; O=$R/gpurun_out/${2:-trace}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 $R/tools/ab_step.py $SW > $O/kt.log 2>&1
cd $R
find $O -name "*kernel_trace.csv" -size +20M -delete
python3 - <<PY
import csv, glob
f = glob.glob("$O/kt/**/p_kernel_stats.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
with open("$O/kernel_stats.txt", "w") as out:
    for r in rows[:45]:
        line = "%9.1f us x %5s  %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], r["Name"][:110])
        out.write(line + "\n")
print(open("$O/kernel_stats.txt").read())
print(open("$O/kt.log").read().strip().splitlines()[-2:])
PY
