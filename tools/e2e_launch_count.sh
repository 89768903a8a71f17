#!/bin/bash
# Runs on the GPU box: kernel launches per propagated frame of examples/propagate_clip.py (eager loop), from two
# rocprofv3 kernel traces that differ only in the clip length.   usage: tools/e2e_launch_count.sh TAG
TAG=$1
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for F in 16 31; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/f$F -o p -- python3 $REPO/examples/propagate_clip.py --frames $F --fused-mask-step > $OUT/f$F.log 2>&1
done
cd $REPO
python3 - $OUT <<'PY'
import csv, glob, sys, re, collections
out = sys.argv[1]
tot = {}
per = {}
for F in (16, 31):
    f = glob.glob("%s/f%d/**/*kernel_stats.csv" % (out, F), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot[F] = sum(int(r["Calls"]) for r in rows)
    per[F] = {re.sub(r"\(.*", "", r["Name"].replace("void ", "").replace("(anonymous namespace)::", ""))[:70]: (int(r["Calls"]), float(r["AverageNs"])) for r in rows}
# the example runs the round twice (warm-up + timed): 2 * (F - 1) propagated frames
frames = 2 * (31 - 16)
print("kernel launches per propagated frame (eager): %.1f" % ((tot[31] - tot[16]) / frames))
rows = []
for k, (c31, ns) in per[31].items():
    c16 = per[16].get(k, (0, 0))[0]
    if c31 != c16:
        rows.append(((c31 - c16) / frames, ns / 1e3, k))
rows.sort(reverse=True)
with open(out + "/per_frame_kernels.csv", "w") as fh:
    fh.write("launches_per_frame,avg_us,kernel\n")
    for n, us, k in rows:
        fh.write("%.2f,%.2f,\"%s\"\n" % (n, us, k))
print(open(out + "/per_frame_kernels.csv").read())
PY
grep -h "frames/s" $OUT/f31.log
rm -rf $OUT/f16 $OUT/f31
