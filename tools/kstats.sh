#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats of one bench.py invocation, top kernels by total time.
#   usage: tools/kstats.sh TAG <bench.py arguments...>
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $REPO/bench.py "$@" > $OUT/log.txt 2>&1
cd $REPO
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:12]:
    print("%-100s %5s calls  avg %8.1f us" % (r["Name"].replace("(anonymous namespace)::", "")[:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -f $OUT/*kernel_trace.csv
