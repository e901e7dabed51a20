#!/usr/bin/env python
"""Idle time of the GPU between consecutive read-queue launches in a rocprofv3 --kernel-trace csv.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --no-cpu-baseline
    python tools/kernel_gaps.py DIR
"""
import csv
import glob
import sys


def main(d):
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    q = [(s, e) for s, e, n in rows if "k_read_queue" in n]
    if len(q) < 3:
        print("too few launches")
        return
    busy = {}
    for s, e, n in rows:
        key = n.split("(")[0].replace("void ", "")[:40]
        busy[key] = busy.get(key, 0) + (e - s)
    gaps = [(q[i + 1][0] - q[i][1]) / 1e6 for i in range(len(q) - 1)]
    durs = [(e - s) / 1e6 for s, e in q]
    print(f"{len(q)} read-queue launches: duration mean {sum(durs)/len(durs):.3f} ms (min {min(durs):.3f} max {max(durs):.3f})")
    print("gap between a launch's end and the next launch's start (ms):", " ".join(f"{g:.2f}" for g in gaps))
    span = (q[-1][1] - q[2][0]) / 1e6
    print(f"steady state (from the 3rd launch): span {span:.1f} ms, queue kernels {sum(durs[2:]):.1f} ms = {sum(durs[2:])/span:.3f} of the time")
    for k, v in sorted(busy.items(), key=lambda x: -x[1])[:8]:
        print(f"   {k:42s} {v/1e6:9.2f} ms total")


if __name__ == "__main__":
    main(sys.argv[1])
