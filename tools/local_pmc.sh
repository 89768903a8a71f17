#!/bin/bash
# Runs on the GPU box: kernel durations + PMC counters of the local-window stage alone (tools/local_bench.py).
# usage: tools/local_pmc.sh TAG
TAG=$1
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- python3 $REPO/tools/local_bench.py 30 > $OUT/stats.log 2>&1
for grp in "wave:GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM" \
           "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${grp%%:*}; ctrs=${grp#*:}
  timeout -k 5 900 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pmc_$name -o p -- python3 $REPO/tools/local_bench.py 6 > $OUT/pmc_$name.log 2>&1
done
cd $REPO
cat $OUT/stats.log | grep "us per call"
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
grep -E "local_fused|lf_pool|frame_prepare" "$f" | cut -d, -f1-4 | cut -c1-150
python3 tools/pmc_summary.py $(find $OUT -path "*pmc_*" -name "*counter_collection.csv" | sort) > $OUT/pmc_summary.csv
grep -E "local_fused|lf_pool|frame_prepare|kernel,counter" $OUT/pmc_summary.csv
rm -rf $OUT/pmc_*/ $OUT/stats
