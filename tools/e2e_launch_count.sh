#!/bin/bash
# Runs on the GPU box: kernel launches per propagated frame of examples/propagate_clip.py (eager loop), from two
# rocprofv3 kernel traces of the SAME clip that differ only in the number of timed interaction rounds (1 vs 3): the
# difference is 2 rounds of propagation -- no encoder, no warm-up, no one-off work.   usage: tools/e2e_launch_count.sh TAG [extra
# propagate_clip.py arguments, e.g. --bank scribble: the strokes-only bank of a session's rounds 2..8]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
F=31
cd /tmp && export TMPDIR=/tmp
for R in 1 3; do
  timeout -k 5 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r$R -o p -- python3 $REPO/examples/propagate_clip.py --frames $F --rounds $R --fused-mask-step "$@" > $OUT/r$R.log 2>&1
done
cd $REPO
python3 - $OUT $F <<'PY'
import csv, glob, sys, re
out, F = sys.argv[1], int(sys.argv[2])
tot, per = {}, {}
for R in (1, 3):
    f = glob.glob("%s/r%d/**/*kernel_stats.csv" % (out, R), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot[R] = sum(int(r["Calls"]) for r in rows)
    per[R] = {re.sub(r"\(.*", "", r["Name"].replace("void ", "").replace("(anonymous namespace)::", ""))[:70]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in rows}
frames = 2 * (F - 1)  # two more rounds of F - 1 propagated frames (each round also runs int_seghead on the annotated frame once)
rows = []
for k, (c3, ns3) in per[3].items():
    c1, ns1 = per[1].get(k, (0, 0.0))
    # (the stand-in encoder's MIOpen kernels run once per process, in find mode: their counts differ from run to run --
    # they are not part of a propagated frame)
    if c3 > c1 and not re.match(r"miopen|Cijk_|naive_conv|igemm_|batched_transpose|Im2d2Col|SubTensorOp|_ZN2ck|MIOpen", k):
        rows.append(((c3 - c1) / frames, (ns3 - ns1) / (c3 - c1) / 1e3, k))
rows.sort(key=lambda r: -r[0] * r[1])
print("kernel launches per propagated frame (eager): %.1f" % sum(r[0] for r in rows))
with open(out + "/per_frame_kernels.csv", "w") as fh:
    fh.write("launches_per_frame,avg_us,us_per_frame,kernel\n")
    for n, us, k in rows:
        fh.write("%.2f,%.2f,%.1f,\"%s\"\n" % (n, us, n * us, k))
    fh.write("total,,%.1f,\"sum of kernel time per propagated frame\"\n" % sum(n * us for n, us, _ in rows))
print(open(out + "/per_frame_kernels.csv").read())
PY
grep -h "frames/s" $OUT/r3.log
rm -rf $OUT/r1 $OUT/r3
