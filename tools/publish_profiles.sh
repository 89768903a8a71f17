#!/bin/bash
# Copies a tools/capture_profiles.sh capture from gpurun_out/SRC into the tracked profiles/ as rNN_* and writes
# the per-config traffic json bench.py reads.   usage: tools/publish_profiles.sh SRC rNN
SRC=$1; R=$2
for c in "2 f32 global_match_f32_pipe_kernel" "3 bf16 global_match_bf16_wide_kernel" "5 bf16 global_match_bf16_wide_kernel"; do
  set -- $c
  t=${SRC}_cfg$1_$2
  for f in bench_line.json bench_line_under_rocprof.json kernel_stats.csv pmc_summary.csv; do
    cp gpurun_out/$SRC/${t}_$f profiles/${R}_cfg$1_$2_$f
  done
  python3 tools/make_traffic_json.py profiles/${R}_cfg$1_$2_pmc_summary.csv $3 $1 $2 "--cfg $1 --compute $2" > profiles/traffic_cfg$1_$2.json
done
ls -la profiles
