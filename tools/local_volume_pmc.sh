#!/bin/bash
# Runs on the GPU box: kernel durations + PMC counters of the local match split at its label boundary (r6) -- the fused kernel, the
# batched phase-1 launch (32 frame pairs per dispatch) and the per-frame phase-2 kernel on stored volumes (tools/local_volume_bench.py:
# loops over 60 rotating frame pairs, i.e. a warm GPU).   usage: tools/local_volume_pmc.sh TAG [bench args...]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- python3 $REPO/tools/local_volume_bench.py "$@" > $OUT/stats.log 2>&1 < /dev/null
for grp in "wave:GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM" \
           "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${grp%%:*}; ctrs=${grp#*:}
  timeout -k 5 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pmc_$name -o p -- python3 $REPO/tools/local_volume_bench.py "$@" --reps 1 > $OUT/pmc_$name.log 2>&1 < /dev/null
done
cd $REPO
grep "us per pair" $OUT/stats.log
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $OUT/kernel_stats.csv; grep -E "local_fused|fill_f32" "$f" | cut -d, -f1-4 | cut -c1-150; fi
python3 tools/pmc_summary.py --last 40 $(find $OUT -path "*pmc_*" -name "*counter_collection.csv" | sort) > $OUT/pmc_summary.csv
grep -E "local_fused|kernel,counter" $OUT/pmc_summary.csv
rm -rf $OUT/pmc_*/ $OUT/stats
