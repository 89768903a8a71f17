#!/bin/bash
# Runs on the GPU box: per-kernel average durations of any python script of this repo.
# usage: tools/kstats_any.sh TAG script.py [args...]     (the program after `--` is python3 itself, never the script)
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
SCRIPT=$REPO/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $SCRIPT "$@" > $OUT/run.log 2>&1
cd $REPO
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    n=re.sub(r"\(anonymous namespace\)::|void ","",r["Name"]); n=re.sub(r"\(.*","",n)
    print("%-64s calls %5s avg %9.2f us  min %9.2f  %5.1f%%" % (n[:64], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["Percentage"])))
PY
tail -12 $OUT/run.log
