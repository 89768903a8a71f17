#!/bin/bash
# Runs on the GPU box.  usage: tools/profile_gpu.sh TAG <bench.py args...>
#   gpurun_out/TAG/stats/   rocprofv3 --kernel-trace --stats (csv)
#   gpurun_out/TAG/pmc_*/   one --pmc pass per counter group (never combined with other trace domains)
# then prints tools/pmc_summary.py tables.  Copy what should be judged into profiles/.
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- python3 $REPO/bench.py "$@" --no-cpu-baseline > $OUT/stats.log 2>&1
for grp in "mfma:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
           "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
           "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${grp%%:*}; ctrs=${grp#*:}
  timeout -k 5 900 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pmc_$name -o p -- python3 $REPO/bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline > $OUT/pmc_$name.log 2>&1
done
cd $REPO
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
echo "== kernel stats ($f)"; head -12 "$f"
python3 tools/pmc_summary.py $(find $OUT -path "*pmc_*" -name "*counter_collection.csv" | sort) > $OUT/pmc_summary.csv
grep -E "global_match|local_|pool|pack_rows|kernel,counter" $OUT/pmc_summary.csv
grep -h '"metric"' $OUT/stats.log | tail -1 > $OUT/bench_line_under_rocprof.json
