# kernel tables of the headline step run eagerly and replayed as a HIP graph, on the same box, one after the other
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/graphkt; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for which in eager_step_only graph_replay_only; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$which -o p -- python3 $R/tools/probes/$which.py > $O/$which.log 2>&1
done
cd $R
python3 - <<PY
import csv, glob
tabs = {}
for which in ("eager_step_only", "graph_replay_only"):
    f = glob.glob("$O/" + which + "/**/p_kernel_stats.csv", recursive=True)[0]
    tabs[which] = {r["Name"]: (float(r["AverageNs"]) / 1e3, int(r["Calls"])) for r in csv.DictReader(open(f))}
    print(which, open("$O/" + which + ".log").read().strip().splitlines()[-1][:80])
names = sorted(tabs["eager_step_only"], key=lambda n: -tabs["eager_step_only"][n][0] * tabs["eager_step_only"][n][1])[:16]
for n in names:
    e = tabs["eager_step_only"][n]
    gname = n if n in tabs["graph_replay_only"] else next((k for k in tabs["graph_replay_only"] if k.split("(")[0] == n.split("(")[0]), None)
    gr = tabs["graph_replay_only"].get(gname, (float("nan"), 0))
    print("%8.1f us eager  %8.1f us graph   %s" % (e[0], gr[0], n[:90]))
PY
