#!/bin/bash
# Runs on the GPU box: kernel durations + fabric traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the head's depthwise
# and 1x1 kernels alone (tools/head_bench.py) on a WARM GPU (r6: 200 launches of warm-up in every pass, the last 10 dispatches of each
# kernel averaged -- r5's passes ran 6 launches right after process start, at ramping clocks).   usage: tools/head_pmc.sh TAG
TAG=$1
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- python3 $REPO/tools/head_bench.py 30 --warm 200 > $OUT/stats.log 2>&1
for grp in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "mfma:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS"; do
  name=${grp%%:*}; ctrs=${grp#*:}
  timeout -k 5 900 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pmc_$name -o p -- python3 $REPO/tools/head_bench.py 12 --warm 200 > $OUT/pmc_$name.log 2>&1
done
cd $REPO
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/kernel_stats.csv
grep -E "dwconv|conv1x1|head_layer1" "$f" | cut -d, -f1-4 | cut -c1-150
python3 tools/pmc_summary.py --last 10 $(find $OUT -path "*pmc_*" -name "*counter_collection.csv" | sort) > $OUT/pmc_summary.csv
grep -E "dwconv|conv1x1|head_layer1|kernel,counter" $OUT/pmc_summary.csv
rm -rf $OUT/pmc_*/ $OUT/stats
