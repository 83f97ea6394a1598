cd /tmp && export TMPDIR=/tmp
timeout 200 python3 $GRAFT_REPO_ROOT/tools/time_flow_fullframe.py 2>&1 | grep -v amdgpu.ids
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ff -o p -- python3 $GRAFT_REPO_ROOT/tools/time_flow_fullframe.py > /dev/null 2>&1
python3 - <<'PY'
import csv,os
rows=list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/ff/p_kernel_stats.csv')))
for r in rows[:10]:
    print(r['Name'][:64], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
