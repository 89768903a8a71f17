#!/bin/bash
# Runs on the GPU box: per-kernel averages of tools/insitu_head.py with and without the global match in the loop.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/insitu; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "0 0" "1 0" "0 1" "1 1"; do set -- $cfg
  timeout -k 5 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/m$1f$2 -o p -- python3 $REPO/tools/insitu_head.py --match $1 --fresh $2 > $OUT/m$1f$2.log 2>&1
  echo "== global match in the loop: $1, fresh inputs: $2"
  python3 - $OUT/m$1f$2 <<'PY'
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(.*", "", r["Name"].replace("void ", "").replace("(anonymous namespace)::", ""))[:50]
    if int(r["Calls"]) >= 100 and re.search("conv1x1|dwconv|global_match_f32", n):
        print("  %-50s calls %5s  avg %8.1f us" % (n, r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $OUT/m$1f$2
done
