#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats (csv) of ONE python command, top kernels by total time -- every step under `timeout`,
# nothing ever reads stdin.   usage: tools/kstat_cmd.sh TAG [filter-regex] -- script.py args...
TAG=$1; shift
FILT=".*"
if [ "$1" != "--" ]; then FILT=$1; shift; fi
shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=/tmp/kstat_$TAG
rm -rf $OUT; mkdir -p $OUT $REPO/gpurun_out
SCRIPT=$1; shift
case "$SCRIPT" in /*) ;; *) SCRIPT=$REPO/$SCRIPT ;; esac
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 "$SCRIPT" "$@" > $REPO/gpurun_out/kstat_$TAG.log 2>&1 < /dev/null
cd $REPO
timeout 60 python3 - $OUT "$FILT" <<'PY' | tee $REPO/gpurun_out/kstat_$TAG.txt
import csv, glob, re, sys
fs = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not fs:
    print("no kernel_stats.csv under", sys.argv[1]); sys.exit(0)
rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: -float(r["TotalDurationNs"]))
pat = re.compile(sys.argv[2])
n = 0
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])
    name = re.sub(r"\(.*", "", name)
    if not pat.search(name):
        continue
    print("%-70s %6s calls  avg %8.2f us  min %8.2f  max %8.2f" % (name[:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
    n += 1
    if n >= 16:
        break
PY
