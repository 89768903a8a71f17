#!/bin/bash
# Copies a tools/capture_profiles.sh capture from gpurun_out/SRC into the tracked profiles/ as rNN_* and writes the
# per-config traffic json bench.py reads -- from the same capture.   usage: tools/publish_profiles.sh SRC rNN
SRC=$1; R=$2
cp gpurun_out/$SRC/${SRC}_bench_line.json profiles/${R}_bench_line.json
cp gpurun_out/$SRC/${SRC}_bench_full.json profiles/${R}_bench_full.json
cp gpurun_out/$SRC/${SRC}_local_pmc_summary.csv profiles/${R}_local_pmc_summary.csv
cp gpurun_out/$SRC/${SRC}_e2e_per_frame_kernels.csv profiles/${R}_e2e_per_frame_kernels.csv
cp gpurun_out/$SRC/${SRC}_head_pointwise.log profiles/${R}_head_pointwise.log
cp gpurun_out/$SRC/${SRC}_frame_prepare.log profiles/${R}_frame_prepare_local.log
for f in head_pmc_summary.csv head_kernel_stats.csv local_volume_pmc_summary.csv local_volume_kernel_stats.csv e2e_per_frame_kernels_scribble_bank.csv; do
  [ -f gpurun_out/$SRC/${SRC}_$f ] && cp gpurun_out/$SRC/${SRC}_$f profiles/${R}_$f
done
for c in "2 f32 global_match_f32_pipe_kernel" "3 bf16 global_match_bf16_wide_kernel<7, 0, false>" "5 bf16 global_match_bf16_wide_kernel<7, 0, false>" "3 bf16r global_match_bf16_wide_kernel<7, 0, true>" "2 bf16r global_match_bf16_wide_kernel<7, 0, true>"; do
  IFS=' ' read -r cfg comp kern <<< "$c"
  t=${SRC}_cfg${cfg}_${comp}
  for f in bench_line_under_rocprof.json kernel_stats.csv pmc_summary.csv; do
    cp gpurun_out/$SRC/${t}_$f profiles/${R}_cfg${cfg}_${comp}_$f
  done
  python3 tools/make_traffic_json.py profiles/${R}_cfg${cfg}_${comp}_pmc_summary.csv "$kern" $cfg $comp "--cfg $cfg --compute $comp" > profiles/traffic_cfg${cfg}_${comp}.json
done
ls -la profiles
