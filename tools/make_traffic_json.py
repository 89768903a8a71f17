#!/usr/bin/env python3
"""profiles/traffic_cfg<N>_<compute>.json from a tools/profile_gpu.sh capture (pmc_summary.csv):
HBM/fabric bytes per launch of the dominant kernel, corrected as MI355X_MICROARCH.md (HBM section) prescribes
for gfx950: FETCH_SIZE (KiB) counts a wide coalesced stream at half its bytes -> x2; WRITE_SIZE as reported.
usage: make_traffic_json.py gpurun_out/TAG/pmc_summary.csv KERNEL_SUBSTRING cfg compute "bench args" > out.json"""
import csv
import datetime
import hashlib
import json
import os
import subprocess
import sys


def main():
    path, kernel, cfg, compute, cmd = sys.argv[1:6]
    fetch = write = dur = None
    name = None
    for r in csv.DictReader(open(path)):
        if kernel not in r["kernel"]:
            continue
        name = r["kernel"]
        if r["counter"] == "FETCH_SIZE":
            fetch, dur = float(r["avg_per_dispatch"]), float(r["avg_duration_us"])
        if r["counter"] == "WRITE_SIZE":
            write = float(r["avg_per_dispatch"])
    assert fetch is not None and write is not None, "FETCH_SIZE / WRITE_SIZE rows of %r not found" % kernel
    try:
        head = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        head = "unknown"
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cvpr2020_manet_amd", "csrc", "global_match.hip")
    json.dump({
        "kernel": name,
        # bench.py compares this with the source of the library it runs: a mismatch marks the figure `stale`
        "kernel_source_sha": hashlib.sha256(open(src, "rb").read()).hexdigest()[:12],
        "cfg": int(cfg), "compute": compute,
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_gpu.sh) of "
                  "`python3 bench.py %s --steps 6 --warmup 2 --no-cpu-baseline`; summary in %s" % (cmd, path),
        "captured_at": "%s (git %s + working tree)" % (datetime.date.today().isoformat(), head),
        "FETCH_SIZE_KiB_per_launch": fetch,
        "WRITE_SIZE_KiB_per_launch": write,
        "kernel_us_under_pmc": dur,
        "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md, "
                      "HBM section) -> fetch bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE uncorrected",
        "hbm_bytes_per_launch": int(2 * fetch * 1024 + write * 1024),
        "note": "memory-side (fabric) requests of the L2s; Infinity-Cache hits are counted.",
    }, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
