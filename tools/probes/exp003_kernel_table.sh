# kernel table of the exp003 train step: rocprofv3 --kernel-trace --stats over tools/time_exp003_only.py 8 6 (8 steps incl. 2 warm-up)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/x3 -o p -- python3 $GRAFT_REPO_ROOT/tools/time_exp003_only.py 8 6 > $GRAFT_REPO_ROOT/gpurun_out/x3.log 2>&1
python3 - <<'PY'
import csv, os
rows = list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/x3/p_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("kernel | calls | average | share of kernel time")
for r in rows[:32]:
    print(f"{r['Name'][:92]:92s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.1f} us {float(r['TotalDurationNs'])/tot*100:6.1f} %")
print(f"total kernel ms per step: {tot/1e6/8:.2f}")
PY
grep -i "ms" $GRAFT_REPO_ROOT/gpurun_out/x3.log | tail -3
