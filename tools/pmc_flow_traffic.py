"""HBM bytes of the advection pipeline per kernel and per batch from two rocprofv3 passes over tools/time_flow_stages.py
(--pmc FETCH_SIZE and, separately, --pmc WRITE_SIZE), corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for
gfx950 (FETCH_SIZE in KB, doubled: it tallies 128-byte requests at 64 bytes; WRITE_SIZE in KB, exact for wide stores).
One batch = the launches of ONE optical_flow.advect_future_frames call (a kernel's launches / batches run).

  python tools/pmc_flow_traffic.py <batch> <fetch dir> <write dir> > profiles/r04/pmc_flow_traffic_B32.json"""
import json
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import collect

BATCHES_RUN = 3 + 30 + 10      # tools/time_flow_stages.py: warm-up + plain + staged


def main():
    b = int(sys.argv[1])
    fetch, nf = collect(sys.argv[2], "FETCH_SIZE")
    write, nw = collect(sys.argv[3], "WRITE_SIZE")
    kernels, total = {}, 0.0
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("pv::"):
            continue
        rd, wr = 2.0 * fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
        n = nf.get(k, nw.get(k, 0))
        kernels[k] = {"launches_per_batch": round(n / BATCHES_RUN, 2), "hbm_read_bytes_per_launch": round(rd / max(n, 1)),
                      "hbm_write_bytes_per_launch": round(wr / max(nw.get(k, 1), 1)),
                      "hbm_GB_per_batch": round((rd + wr) / BATCHES_RUN / 1e9, 4)}
        total += (rd + wr) / BATCHES_RUN
    json.dump({"batch": b, "GB_per_batch": round(total / 1e9, 4), "batches_run": BATCHES_RUN,
               "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) -- python3 tools/time_flow_stages.py 32",
               "units": "hbm_read = 2 * FETCH_SIZE[KB] * 1024 (gfx950 correction), hbm_write = WRITE_SIZE[KB] * 1024", "kernels": kernels},
              sys.stdout, indent=1)


if __name__ == "__main__":
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    main()
