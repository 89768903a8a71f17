#!/bin/bash
# Runs on the GPU box: the round's tracked evidence.  usage: tools/capture_profiles.sh RTAG   (e.g. r02)
#   for each of: cfg2 f32 (headline), cfg3 bf16, cfg5 bf16
#     profiles/RTAG_cfgN_<compute>_bench_line.json        un-profiled `python bench.py ...` line
#     gpurun_out/RTAG_cfgN_<compute>/...                  rocprofv3 kernel stats + PMC passes (tools/profile_gpu.sh)
# The caller copies the summaries from gpurun_out/ into profiles/ (gpurun merges only gpurun_out/ back).
R=$1
mkdir -p gpurun_out/$R
for c in "2 f32" "3 bf16" "5 bf16"; do
  set -- $c
  tag=${R}_cfg$1_$2
  if [ "$1" = "2" ]; then
    python bench.py --cfg $1 --compute $2 --steps 40 > gpurun_out/$R/${tag}_bench_line.json 2> gpurun_out/$R/${tag}_bench.err
  else
    python bench.py --cfg $1 --compute $2 --steps 40 > gpurun_out/$R/${tag}_bench_line.json 2> gpurun_out/$R/${tag}_bench.err
  fi
  tools/profile_gpu.sh ${R}/$tag --cfg $1 --compute $2 --steps 20 > gpurun_out/$R/${tag}_profile.log 2>&1
  cp gpurun_out/${R}/$tag/pmc_summary.csv gpurun_out/$R/${tag}_pmc_summary.csv
  cp $(find gpurun_out/${R}/$tag/stats -name "*kernel_stats.csv" | head -1) gpurun_out/$R/${tag}_kernel_stats.csv
  cp gpurun_out/${R}/$tag/bench_line_under_rocprof.json gpurun_out/$R/${tag}_bench_line_under_rocprof.json
  rm -rf gpurun_out/${R}/$tag/stats gpurun_out/${R}/$tag/pmc_*   # keep the merge small
  tail -c 600 gpurun_out/$R/${tag}_bench_line.json | head -c 300; echo
done
