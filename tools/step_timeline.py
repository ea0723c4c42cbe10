#!/usr/bin/env python3
"""Print the kernel timeline of the last two training steps of a `rocprofv3 --kernel-trace --output-format csv` run of
tools/train_bench.py (start offset in us, duration in us, queue, kernel) -- where the GPU idles between launches.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 tools/train_bench.py --steps 30
    python3 tools/step_timeline.py $(find /tmp/tr -name '*kernel_trace.csv') > gpurun_out/timeline.txt
"""
import csv
import sys


def main(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "adam_step_kernel" in r["Kernel_Name"] or "FusedAdam" in r["Kernel_Name"]]
    a, b = marks[-3], marks[-1]
    t0 = int(rows[a]["Start_Timestamp"])
    busy, last_end = 0, t0
    for r in rows[a:b + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("void ", "")[:72]
        print("%9.1f %8.1f q%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), name))
        if e > last_end:
            busy += e - max(s, last_end)
            last_end = e
    span = int(rows[b]["End_Timestamp"]) - t0
    print("# span %.1f us, GPU busy %.1f us (%.0f %%)" % (span / 1e3, busy / 1e3, 100.0 * busy / span))


if __name__ == "__main__":
    main(sys.argv[1])
