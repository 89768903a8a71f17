#!/bin/bash
# Runs on the GPU box: the bf16r step on the data kinds for the product library and several variants (same box).
#   usage: tools/ab_many.sh "kinds" variant1.so variant2.so ...
KINDS=$1; shift
one() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms/step, filter %.3f ms' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for d in $KINDS; do
  echo "$d product: $(python bench.py --data $d --compute bf16r --emb f32 --steps 30 --no-also --no-robustness --no-e2e --no-cpu-baseline 2>/dev/null | one)"
  for V in "$@"; do
    echo "$d $(basename $V): $(MANET_LIB_VARIANT=$V python tools/bench_variant.py --data $d --compute bf16r --emb f32 --steps 30 --no-also --no-robustness --no-e2e --no-cpu-baseline 2>/dev/null | one)"
  done
done
