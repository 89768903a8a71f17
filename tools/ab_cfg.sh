#!/bin/bash
# Runs on the GPU box: A/B of one bench configuration, product library vs a variant (same box, interleaved).
#   usage: tools/ab_cfg.sh path/to/variant.so <bench.py arguments...>
V=$1; shift
one() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms/step, main kernel %.3f ms' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for rep in 1 2 3; do
  echo "product: $(python bench.py "$@" --no-also --no-robustness --no-e2e --no-cpu-baseline 2>/dev/null | one)"
  echo "variant: $(MANET_LIB_VARIANT=$V python tools/bench_variant.py "$@" --no-also --no-robustness --no-e2e --no-cpu-baseline 2>/dev/null | one)"
done
