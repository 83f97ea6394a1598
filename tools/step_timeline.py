"""Per-step kernel timeline of the headline train step from a rocprofv3 --kernel-trace CSV:
   rocprofv3 --kernel-trace --output-format csv -d OUT -o p -- python3 bench.py --steps 10 --warmup 3 --no-roofline --no-cpu-baseline
   python tools/step_timeline.py OUT [n_steps]
Prints, for the LAST step, every launch with its start offset, duration and the idle gap in front of it, and the
sums over the last n_steps (busy time vs wall time)."""
import csv
import glob
import re
import sys

path = sys.argv[1]
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows = []
for f in glob.glob(f"{path}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")))
rows.sort()
# a step starts at the first launch of the satellite normalise / pack kernel; use the fc1 fused update as the step end marker
ends = [i for i, r in enumerate(rows) if "linear_bwd_dw_" in r[2]]
ends = ends[-(n_steps + 1):]
tot_busy = tot_wall = 0
for a, b in zip(ends[:-1], ends[1:]):
    seg = rows[a + 1:b + 1]
    tot_busy += sum(e - s for s, e, _ in seg)
    tot_wall += seg[-1][1] - rows[a][1]
print(f"last {len(ends) - 1} steps: wall {tot_wall / (len(ends) - 1) / 1e3:.1f} us/step, kernel busy {tot_busy / (len(ends) - 1) / 1e3:.1f} us/step, launches/step {ends[-1] - ends[-2]}")
a, b = ends[-2], ends[-1]
prev_end = rows[a][1]
t0 = prev_end
for s, e, name in rows[a + 1:b + 1]:
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:6.1f}  {name[:100]}")
    prev_end = max(prev_end, e)
