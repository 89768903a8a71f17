#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (``--kernel-trace --stats``) as a small CSV:
kernel name (shortened), calls, total us, average us, percentage.   usage: rocpd_summary.py x.db"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:<>, ]+?)\(", name)
    base = m.group(1) if m else name
    return base[:90]


def main(path):
    db = sqlite3.connect(path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print("kernel,calls,total_us,avg_us,percent")
    for name, calls, total, avg, pct in rows:
        print("%s,%d,%.1f,%.2f,%.2f" % (short(name).replace(",", ";"), calls, total, avg, pct))


if __name__ == "__main__":
    main(sys.argv[1])
