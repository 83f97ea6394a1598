"""Aggregates two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; collected separately, `--pmc` only) into
per-kernel HBM bytes per launch, corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950
(FETCH_SIZE counts KB and reports half of the bytes of wide coalesced reads; WRITE_SIZE is exact, KB).

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline
  python tools/pmc_traffic.py $OUT/fetch $OUT/write > profiles/rNN/pmc_hbm_traffic_bench_B32.json
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)             # drop the argument list
    return name.replace("void ", "").strip()


def collect(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(f"{path}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            tot[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return tot, cnt


def main():
    fetch, nf = collect(sys.argv[1], "FETCH_SIZE")
    write, nw = collect(sys.argv[2], "WRITE_SIZE")
    out = {"command": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) -- python3 bench.py --steps 3 --warmup 1 "
                      "--no-cpu-baseline --no-roofline (B=32, T=18)",
           "units": "bytes per launch; hbm_read_bytes = 2 * FETCH_SIZE[KB] * 1024 (gfx950 correction of MI355X_MICROARCH.md), "
                    "hbm_write_bytes = WRITE_SIZE[KB] * 1024",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("pv::"):
            continue
        rd = 2.0 * fetch.get(k, 0.0) * 1024 / max(nf.get(k, 1), 1)
        wr = write.get(k, 0.0) * 1024 / max(nw.get(k, 1), 1)
        out["kernels"][k] = {"launches": nf.get(k, nw.get(k, 0)), "hbm_read_bytes_per_launch": round(rd),
                             "hbm_write_bytes_per_launch": round(wr), "hbm_bytes_per_launch": round(rd + wr)}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
