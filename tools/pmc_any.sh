#!/bin/bash
# Runs on the GPU box: kernel stats + PMC counter groups for any python script.   usage: tools/pmc_any.sh TAG script.py [args]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- python3 $REPO/"$@" > $OUT/stats.log 2>&1
for grp in "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_VMEM" "wave:GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM" \
           "fetch:FETCH_SIZE" "write:WRITE_SIZE" "l2:TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  name=${grp%%:*}; ctrs=${grp#*:}
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pmc_$name -o p -- python3 $REPO/"$@" > $OUT/pmc_$name.log 2>&1
done
cd $REPO
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
head -8 "$f" | cut -d, -f1-4 | cut -c1-160
python3 tools/pmc_summary.py $(find $OUT -path "*pmc_*" -name "*counter_collection.csv" | sort) > $OUT/pmc_summary.csv
cat $OUT/pmc_summary.csv | grep -v "^at::\|elementwise\|fill"
rm -rf $OUT/pmc_*/ $OUT/stats
