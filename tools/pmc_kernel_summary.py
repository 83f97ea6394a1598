"""Sums rocprofv3 --pmc counter passes per kernel (one directory per pass; counters of different passes are merged) and
derives the ratios DESIGN.md quotes.

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \\
            --output-format csv -d $OUT/a -o p -- python3 tools/run_conv_only.py fwd
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \\
            --output-format csv -d $OUT/b -o p -- python3 tools/run_conv_only.py fwd
  python tools/pmc_kernel_summary.py <kernel-name-substring> <waves per SIMD> $OUT/a $OUT/b [...] > profiles/rNN/pmc_<kernel>.json

Units on gfx950 (checked against instruction counts: SQ_INSTS_MFMA x 16 or 32 cycles = SQ_VALU_MFMA_BUSY_CYCLES exactly):
SQ_VALU_MFMA_BUSY_CYCLES counts SIMD clock cycles; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count per wave in units
of 4 cycles.  "matrix pipe busy while resident" = MFMA busy cycles / (SIMD-resident cycles) with SIMD-resident cycles =
4 x SQ_WAVE_CYCLES / (waves sharing a SIMD).
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def main():
    want = sys.argv[1]
    waves_per_simd = int(sys.argv[2])
    tot, cnt = defaultdict(float), defaultdict(int)
    name = None
    for path in sys.argv[3:]:
        for f in glob.glob(f"{path}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
                if want not in k:
                    continue
                name = k
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[r["Counter_Name"]] += 1
    per = {c: tot[c] / cnt[c] for c in sorted(tot)}
    out = {"kernel": name, "dispatches_averaged": max(cnt.values()) if cnt else 0, "counters_per_dispatch": per, "derived": {}}
    d = out["derived"]
    g = per.get
    if g("SQ_WAVE_CYCLES") and g("SQ_VALU_MFMA_BUSY_CYCLES"):
        d["waves_per_simd"] = waves_per_simd
        d["matrix_pipe_busy_while_resident"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") * waves_per_simd / (4 * g("SQ_WAVE_CYCLES")), 4)
    if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_ANY"):
        d["wait_any_over_wave_cycles"] = round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 4)
    if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_INST_ANY"):
        d["wait_inst_any_over_wave_cycles"] = round(g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), 4)
    if g("SQ_WAVE_CYCLES") and g("SQ_ACTIVE_INST_ANY"):
        d["issuing_over_wave_cycles"] = round(g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"), 4)
    if g("SQ_LDS_IDX_ACTIVE") and g("SQ_LDS_BANK_CONFLICT") is not None:
        d["lds_bank_conflict_over_idx_active"] = round(g("SQ_LDS_BANK_CONFLICT", 0.0) / g("SQ_LDS_IDX_ACTIVE"), 4)
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
