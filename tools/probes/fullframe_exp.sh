cd /tmp && export TMPDIR=/tmp
for e in 0 1 2; do
export PV_FB_EXP=$e
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ffe$e -o p -- python3 $GRAFT_REPO_ROOT/tools/time_flow_fullframe.py > /dev/null 2>&1
python3 - <<PY
import csv,os
rows=list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/ffe$e/p_kernel_stats.csv')))
for r in rows[:3]:
    print($e, r['Name'][:64], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
done
