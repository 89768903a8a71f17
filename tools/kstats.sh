#!/bin/bash
# Runs on the GPU box: per-kernel average durations of one bench configuration.
# usage: tools/kstats.sh TAG <bench.py args...>
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $REPO/bench.py "$@" --no-cpu-baseline > $OUT/run.log 2>&1
cd $REPO
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    n=re.sub(r"\(anonymous namespace\)::|void ","",r["Name"]); n=re.sub(r"\(.*","",n)
    print("%-58s calls %4s avg %9.2f us  %5.1f%%" % (n[:58], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
grep -h '"metric"' $OUT/run.log | tail -1 | python3 -c "import sys,json; l=json.loads(sys.stdin.read()); print('bench:', round(l['value'],1),'fps', round(l['ms_per_step'],3),'ms/step')"
