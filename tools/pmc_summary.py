#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters (counter_collection.csv, one pass per file).
usage: pmc_summary.py dir1/p_counter_collection.csv [dir2/...]   -> CSV on stdout"""
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
    out = csv.writer(sys.stdout)
    out.writerow(["kernel", "counter", "avg_per_dispatch", "dispatches", "avg_duration_us"])
    for p in paths:
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"])
            if k.startswith("at::") or "rocclr" in k:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k in sorted(acc):
            for c in sorted(acc[k]):
                v = acc[k][c]
                out.writerow([k, c, "%.4f" % (sum(v) / len(v)), len(v), "%.2f" % (sum(dur[k]) / len(dur[k]))])


if __name__ == "__main__":
    main(sys.argv[1:])
