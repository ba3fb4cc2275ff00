#!/usr/bin/env python3
"""Condense rocprofv3 output directories into the small files kept under profiles/.

usage: summarize_rocprof.py <dir with stats/ pmc_*/ sub-directories> <profiles/rN>

  stats/      rocprofv3 --kernel-trace --stats --output-format csv   -> kernel_stats.csv
  pmc_*/      rocprofv3 --kernel-trace --pmc <counters> (one pass per directory)
                                                                     -> pmc_summary.json
Counter values are summed over the dispatch's per-XCD/per-instance rows exactly as
rocprofv3 writes them and averaged over the launches of each kernel.  FETCH_SIZE /
WRITE_SIZE are reported by the hardware in KiB-like units of 1 KB and FETCH_SIZE has to be
doubled on gfx950 (MI355X_MICROARCH.md, HBM/rocprofv3 section); both corrections are applied
in the derived "hbm_bytes" block, never in the raw means.
"""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return name if len(name) < 160 else name[:157] + "..."


def main(src, dst):
    os.makedirs(dst, exist_ok=True)
    st = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))
    if st:
        with open(st[0]) as f, open(os.path.join(dst, "kernel_stats.csv"), "w") as g:
            r = csv.reader(f)
            w = csv.writer(g)
            for row in r:
                row[0] = short(row[0])
                w.writerow(row)
    acc = defaultdict(lambda: defaultdict(list))
    for cc in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
        per = defaultdict(float)
        with open(cc) as f:
            for row in csv.DictReader(f):
                key = (row["Dispatch_Id"], short(row["Kernel_Name"]), row["Counter_Name"])
                per[key] += float(row["Counter_Value"])
        for (_, k, c), v in per.items():
            acc[k][c].append(v)
    out = {}
    for k, cs in sorted(acc.items()):
        if k.startswith("void at::") or k.startswith("__amd"):
            continue
        out[k] = {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in sorted(cs.items())}
        d = out[k]
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            fb = d["FETCH_SIZE"]["mean"] * 1024 * 2
            wb = d["WRITE_SIZE"]["mean"] * 1024
            d["hbm_bytes"] = {"fetch_corrected": fb, "write": wb, "total": fb + wb}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d and d["GRBM_GUI_ACTIVE"]["mean"]:
            # MFMA-busy cycles summed over the 1024 SIMDs / (kernel cycles x 1024 SIMDs);
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs, so kernel cycles = GRBM / 8
            d["mfma_busy_frac"] = d["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / (d["GRBM_GUI_ACTIVE"]["mean"] / 8 * 1024)
        if "TCC_HIT_sum" in d and "TCC_REQ_sum" in d and d["TCC_REQ_sum"]["mean"]:
            d["l2_hit_rate"] = d["TCC_HIT_sum"]["mean"] / d["TCC_REQ_sum"]["mean"]
    with open(os.path.join(dst, "pmc_summary.json"), "w") as f:
        json.dump(out, f, indent=1)
    for lg in glob.glob(os.path.join(src, "*.log")):
        with open(lg) as f:
            lines = [l for l in f if l.startswith("{")]
        if lines:
            with open(os.path.join(dst, "bench_" + os.path.basename(lg)[:-4] + ".json"), "w") as g:
                g.write(lines[-1])
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk in ("hbm_bytes", "mfma_busy_frac", "l2_hit_rate")}
                      for k, v in out.items()}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
