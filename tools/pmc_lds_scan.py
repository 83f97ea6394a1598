"""Per-kernel LDS bank-conflict scan from one rocprofv3 --pmc pass:
   rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d OUT -o p -- python3 <workload>
   python tools/pmc_lds_scan.py OUT"""
import csv, glob, re, sys
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
for f in glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            cnt[k] += 1
rows = []
for k, c in tot.items():
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0:
        continue
    rows.append((wc, k, cnt[k], c.get("SQ_LDS_IDX_ACTIVE", 0) / wc, c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0), 1.0),
                 c.get("SQ_WAIT_ANY", 0) / wc))
rows.sort(reverse=True)
print(f"{'kernel':70s} {'calls':>6s} {'wave-cycles share':>8s} {'LDS active / wave cycles':>10s} {'conflict / LDS active':>10s} {'wait_any':>8s}")
allwc = sum(r[0] for r in rows)
for wc, k, n, la, cf, wa in rows[:30]:
    print(f"{k[:70]:70s} {n:6d} {wc / allwc:8.3f} {la:10.3f} {cf:10.3f} {wa:8.3f}")
