#!/bin/bash
# Runs on the GPU box: the round's tracked evidence.  usage: tools/capture_profiles.sh RTAG   (e.g. r03)
#   gpurun_out/RTAG/RTAG_bench_line.json                 the default `python bench.py` compact line (what the driver parses);
#   gpurun_out/RTAG/RTAG_bench_full.json                 ... and the full blocks of the same run (the `#bench_full` stdout line)
#   for each of: cfg2 f32 (headline), cfg3 bf16, cfg5 bf16, cfg3 bf16r and cfg2 bf16r (fp32-stored embeddings)
#     gpurun_out/RTAG/RTAG_cfgN_<compute>_{bench_line_under_rocprof.json,kernel_stats.csv,pmc_summary.csv}
#                                                        rocprofv3 kernel stats + separate --pmc passes (tools/profile_gpu.sh)
#   gpurun_out/RTAG/RTAG_local_pmc_summary.csv           the local-window stage alone (tools/local_pmc.sh)
# tools/publish_profiles.sh copies them into the tracked profiles/ and regenerates profiles/traffic_cfg*.json from the SAME
# capture (the r2 bench lines quoted an older capture's traffic).
R=$1
mkdir -p gpurun_out/$R
python bench.py > gpurun_out/$R/${R}_bench_stdout.txt 2> gpurun_out/$R/${R}_bench.err
tail -n 1 gpurun_out/$R/${R}_bench_stdout.txt > gpurun_out/$R/${R}_bench_line.json                          # the compact driver line
grep -h '^#bench_full ' gpurun_out/$R/${R}_bench_stdout.txt | sed 's/^#bench_full //' > gpurun_out/$R/${R}_bench_full.json  # every block
for c in "2 f32 auto" "3 bf16 auto" "5 bf16 auto" "3 bf16r f32" "2 bf16r f32"; do
  set -- $c
  tag=${R}_cfg$1_$2
  tools/profile_gpu.sh ${R}/$tag --cfg $1 --compute $2 --emb $3 --steps 20 --no-also --no-robustness --no-e2e > gpurun_out/$R/${tag}_profile.log 2>&1
  cp gpurun_out/${R}/$tag/pmc_summary.csv gpurun_out/$R/${tag}_pmc_summary.csv
  cp $(find gpurun_out/${R}/$tag/stats -name "*kernel_stats.csv" | head -1) gpurun_out/$R/${tag}_kernel_stats.csv
  cp gpurun_out/${R}/$tag/bench_line_under_rocprof.json gpurun_out/$R/${tag}_bench_line_under_rocprof.json
  rm -rf gpurun_out/${R}/$tag/stats gpurun_out/${R}/$tag/pmc_*   # keep the merge small
  head -c 300 gpurun_out/$R/${tag}_bench_line_under_rocprof.json; echo
done
tools/local_pmc.sh $R/local > gpurun_out/$R/${R}_local.log 2>&1
cp gpurun_out/$R/local/pmc_summary.csv gpurun_out/$R/${R}_local_pmc_summary.csv
tools/e2e_launch_count.sh $R/e2e > gpurun_out/$R/${R}_e2e.log 2>&1
cp gpurun_out/$R/e2e/per_frame_kernels.csv gpurun_out/$R/${R}_e2e_per_frame_kernels.csv
tail -3 gpurun_out/$R/${R}_e2e.log
tools/e2e_launch_count.sh $R/e2e_scribble --bank scribble > gpurun_out/$R/${R}_e2e_scribble.log 2>&1   # a session's rounds 2..8
cp gpurun_out/$R/e2e_scribble/per_frame_kernels.csv gpurun_out/$R/${R}_e2e_per_frame_kernels_scribble_bank.csv
python tools/pw_bench.py > gpurun_out/$R/${R}_head_pointwise.log 2>&1
python tools/pw_rw_bench.py >> gpurun_out/$R/${R}_head_pointwise.log 2>&1
tail -8 gpurun_out/$R/${R}_head_pointwise.log
python tools/frame_prep_bench.py > gpurun_out/$R/${R}_frame_prepare.log 2>&1
python tools/local_kernel_us.py >> gpurun_out/$R/${R}_frame_prepare.log 2>&1
tail -12 gpurun_out/$R/${R}_frame_prepare.log
# r6: the head's kernels on a WARM GPU and the local match split at its label boundary (stored volumes)
tools/head_pmc.sh $R/head > gpurun_out/$R/${R}_head.log 2>&1
cp gpurun_out/$R/head/pmc_summary.csv gpurun_out/$R/${R}_head_pmc_summary.csv
cp gpurun_out/$R/head/kernel_stats.csv gpurun_out/$R/${R}_head_kernel_stats.csv
tail -4 gpurun_out/$R/${R}_head.log
tools/local_volume_pmc.sh $R/localvol --d 12 > gpurun_out/$R/${R}_localvol.log 2>&1
cp gpurun_out/$R/localvol/pmc_summary.csv gpurun_out/$R/${R}_local_volume_pmc_summary.csv
cp gpurun_out/$R/localvol/kernel_stats.csv gpurun_out/$R/${R}_local_volume_kernel_stats.csv
head -12 gpurun_out/$R/${R}_localvol.log
