#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters (counter_collection.csv, one pass per file).
usage: pmc_summary.py [--last N] dir1/p_counter_collection.csv [dir2/...]   -> CSV on stdout
--last N: only the last N dispatches of every kernel in a file count (a capture that warms the GPU up first: the early launches run at
ramping clocks)"""
import collections
import csv
import re
import sys

csv.field_size_limit(1 << 30)


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"\s*([A-Za-z0-9_:]+(<[0-9a-z_, ]+>)?)", name)
    return (m.group(1) if m else name)[:60]


def main(paths):
    last = 0
    if paths and paths[0] == "--last":
        last, paths = int(paths[1]), paths[2:]
    out = csv.writer(sys.stdout)
    out.writerow(["kernel", "counter", "avg_per_dispatch", "dispatches", "avg_duration_us"])
    for p in paths:
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        dur = collections.defaultdict(list)
        rows = list(csv.DictReader(open(p)))
        keep = None
        if last > 0:  # the last N dispatch ids of every kernel
            ids = collections.defaultdict(list)
            for r in rows:
                k, d = short(r["Kernel_Name"]), int(r["Dispatch_Id"])
                if not ids[k] or ids[k][-1] != d:
                    ids[k].append(d)
            keep = {k: set(sorted(set(v))[-last:]) for k, v in ids.items()}
        for r in rows:
            k = short(r["Kernel_Name"])
            if k.startswith("at::") or "rocclr" in k:
                continue
            if keep is not None and int(r["Dispatch_Id"]) not in keep[k]:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k in sorted(acc):
            for c in sorted(acc[k]):
                v = acc[k][c]
                out.writerow([k, c, "%.4f" % (sum(v) / len(v)), len(v), "%.2f" % (sum(dur[k]) / len(dur[k]))])


if __name__ == "__main__":
    main(sys.argv[1:])
