#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (gpurun_out/pmc_*/.../*_counter_collection.csv)
into one JSON: per kernel and grid size, the mean counter value per launch.

    python tools/pmc_summary.py gpurun_out profiles/r01_pmc_summary.json

HBM traffic per launch follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are in KiB and come from separate passes; on gfx950 FETCH_SIZE counts
128-B requests of wide (16 B/lane) streaming reads at 64 B, so the read side is
doubled (our reads are LDS-DMA dwordx4 and dword loads; the factor is an upper
bound for the latter)."""
import collections
import csv
import glob
import json
import os
import sys


def main(src, dst):
    out = collections.defaultdict(lambda: collections.defaultdict(dict))
    for path in sorted(glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv"))):
        agg = collections.defaultdict(lambda: [0, 0.0])
        with open(path) as f:
            for r in csv.DictReader(f):
                name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                k = (name, r["Grid_Size"], r["Counter_Name"])
                agg[k][0] += 1
                agg[k][1] += float(r["Counter_Value"])
        for (name, grid, ctr), (n, v) in agg.items():
            out[name][grid][ctr] = {"launches": n, "mean_per_launch": v / n}
    for name, grids in out.items():
        for grid, c in grids.items():
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                rd = c["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2
                wr = c["WRITE_SIZE"]["mean_per_launch"] * 1024
                c["hbm_bytes_per_launch"] = {"read_corrected_x2": rd, "write": wr, "total": rd + wr}
    # which build of the kernels these counters belong to: bench.py reports `roofline.traffic` from this file only
    # while the kernel sources still hash to the same value
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_hash
    out["_meta"] = {"csrc_hash": csrc_hash(), "source": "rocprofv3 --pmc passes under " + src}
    with open(dst, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", dst)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
