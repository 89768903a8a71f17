#!/usr/bin/env python3
"""prints the `robustness` block of a bench.py run as a table: file argument = bench_full.json, or a capture of bench.py's
stdout (the `#bench_full ` line is read; the compact last line carries only the block's summary)"""
import json
import sys

lines = open(sys.argv[1]).read().strip().splitlines()
full = [l[len("#bench_full "):] for l in lines if l.startswith("#bench_full ")]
d = json.loads(full[-1] if full else lines[-1])
rb = d["robustness"]
print(rb["summary"])
for leg in rb["legs"]:
    print("%-7s scale %.1f %-6s %8.1f frames/s %7.3f ms/step  kernel %.3f ms  err %.2e  rows/pair %s  rescued %s  skipped filter %s  bit-equal %s"
          % (leg["data"], leg["scale"], leg["compute"], leg["frames_per_s"], leg["ms_per_step"], leg["main_kernel_ms"],
             leg["err_vs_fp32_oracle_normalised_max"], leg.get("candidate_rows_per_pair"), leg.get("rescued_tile_fraction"),
             leg.get("timed_frames_skipped_the_filter"), leg.get("bit_equal_to_f32")))
