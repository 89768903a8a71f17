#!/bin/bash
# Runs on the GPU box: A/B of the bf16r step on the data kinds, product library vs a variant (same box, interleaved).
#   usage: tools/ab_filter.sh path/to/variant.so [kinds...]
V=$1; shift
KINDS=${@:-iid video smooth}
one() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms/step, filter %.3f ms' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for d in $KINDS; do
  for rep in 1 2; do
    echo "$d product: $(python bench.py --data $d --compute bf16r --emb f32 --steps 30 --no-also --no-robustness --no-e2e --no-cpu-baseline 2>/dev/null | one)"
    echo "$d variant: $(MANET_LIB_VARIANT=$V python tools/bench_variant.py --data $d --compute bf16r --emb f32 --steps 30 --no-also --no-robustness --no-e2e --no-cpu-baseline 2>/dev/null | one)"
  done
done
