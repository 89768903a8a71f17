#!/bin/bash
# Runs on the GPU box: the kernel / memory-copy sequence of the last propagated frames of examples/propagate_clip.py (eager loop)
# with the idle gap in front of each launch -- how the blocking host-to-device copy behind `table[i][j] = weight` was found (r5).
mkdir -p gpurun_out/seq
cd /tmp && export TMPDIR=/tmp
timeout -k 5 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/seq/t -o p -- python3 $GRAFT_REPO_ROOT/examples/propagate_clip.py --frames 9 --rounds 2 --fused-mask-step > $GRAFT_REPO_ROOT/gpurun_out/seq/log.txt 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, re
f = glob.glob("gpurun_out/seq/t/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
m = glob.glob("gpurun_out/seq/t/**/*memory_copy_trace.csv", recursive=True)
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))[:60]) for r in rows]
if m:
    for r in csv.DictReader(open(m[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "MEMCPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
# last 45 events
prev_end = None
for s, e, k in ev[-60:]:
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print("%8.1f us gap %6.1f  dur %7.1f  %s" % ((s - ev[-60][0]) / 1e3, gap, (e - s) / 1e3, k))
    prev_end = e
PY
rm -rf gpurun_out/seq/t
