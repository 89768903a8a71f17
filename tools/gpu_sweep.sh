#!/bin/bash
# usage: tools/gpu_sweep.sh TAG "cfg compute tune" ...   (runs on the GPU box; prints one line per run)
TAG=$1; shift
mkdir -p gpurun_out/$TAG
for c in "$@"; do
  set -- $c
  f=gpurun_out/$TAG/bench_cfg$1_$2_t$(echo ${3:-none} | tr '=,' '__').json
  timeout 300 python bench.py --cfg $1 --compute $2 --steps 30 --no-cpu-baseline ${3:+--tune $3} > $f 2> ${f%.json}.err
  python - "$f" <<'PY'
import json,sys
try:
    l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(l["value"],1), "fps", round(l["ms_per_step"],3), "ms/step kern_ms", round(l["roofline"]["kernel_ms"],4), "frac", round(l["roofline"]["frac"],3))
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(sys.argv[1][:-5]+".err").read()[-1500:])
PY
done
