#!/bin/bash
# Makes the first multi-GPU run boring: exercises, on whatever GPUs this box has, every piece an
# N-rank run of bench.py / examples/propagate_clip.py depends on, and prints the `collective` blocks.
#   usage: tools/preflight_multigpu.sh [--dry-run] [--gpus N]      (N default: min(2, visible GPUs); run from the repo root)
#   1. RCCL with ONE rank (MANET_BENCH_FORCE_DIST=1): process-group init with device_id, the all-gather code path
#   2. bench.py --gpus N          (weak scaling; RCCL when N <= visible GPUs, else gloo with ranks sharing a device)
#   3. bench.py --gpus N --scaling strong
#   4. bench.py --e2e --gpus N    (clip-parallel real propagation: all-gather of the clip + one gather per round)
# Environment the N-rank runs depend on:
#   HSA_ENABLE_IPC_MODE_LEGACY=0   dmabuf IPC only on these hosts: RCCL's device-buffer exchange fails without it
#                                  (bench.py / propagate_clip.py set it for the ranks they spawn; exported here for launchers)
#   MASTER_ADDR=127.0.0.1          the container hostname may not resolve (both scripts pass --master-addr 127.0.0.1)
#   MANET_BENCH_BACKEND=gloo       dry run of the N-rank flow on fewer GPUs than ranks (collectives staged through the host)
#   MANET_BENCH_FORCE_DIST=1       take the collective code path with one rank
#   RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT   set by torch.distributed.run
DRY=0; N=""
while [ $# -gt 0 ]; do
  case "$1" in
    --dry-run) DRY=1;;
    --gpus) shift; N=$1;;
    *) echo "usage: tools/preflight_multigpu.sh [--dry-run] [--gpus N]" >&2; exit 2;;
  esac
  shift
done
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
export MASTER_ADDR=127.0.0.1
if [ $DRY -eq 1 ]; then VIS=${PREFLIGHT_VISIBLE_GPUS:-1}; else VIS=$(python3 -c "import torch; print(torch.cuda.device_count())"); fi
[ -z "$N" ] && N=2
BACKEND_ENV=""
[ "$N" -gt "$VIS" ] && BACKEND_ENV="MANET_BENCH_BACKEND=gloo"
LEAN="--steps 6 --warmup 2 --no-cpu-baseline --no-also --no-robustness --no-e2e"
OUT=gpurun_out/preflight; mkdir -p $OUT
run() {  # name, env assignments (may be empty), command...
  local name=$1 envs=$2; shift 2
  echo "== $name: ${envs:+$envs }$*"
  [ $DRY -eq 1 ] && return 0
  env $envs "$@" > $OUT/$name.json 2> $OUT/$name.err
  local rc=$?
  python3 - "$OUT/$name.json" $rc <<'PY'
import json, sys
try:
    l = json.loads([x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")][-1])
    print("   rc %s  n_gpus %s  value %.1f %s  collective %s" % (sys.argv[2], l.get("n_gpus"), l.get("value", 0.0), l.get("unit"), json.dumps(l.get("collective"))))
except Exception as e:
    print("   rc %s  NO JSON LINE (%s) -- see %s" % (sys.argv[2], e, sys.argv[1].replace(".json", ".err")))
PY
  return $rc
}
echo "visible GPUs: $VIS, ranks: $N${BACKEND_ENV:+ ($BACKEND_ENV: ranks share devices)}, HSA_ENABLE_IPC_MODE_LEGACY=$HSA_ENABLE_IPC_MODE_LEGACY"
FAIL=0
run rccl_one_rank "MANET_BENCH_FORCE_DIST=1" python3 bench.py $LEAN || FAIL=1
run weak_n$N "$BACKEND_ENV" python3 bench.py --gpus $N $LEAN || FAIL=1
run strong_n$N "$BACKEND_ENV" python3 bench.py --gpus $N --scaling strong $LEAN || FAIL=1
run e2e_n$N "$BACKEND_ENV" python3 bench.py --e2e --gpus $N --e2e-frames 16 || FAIL=1
[ $DRY -eq 1 ] && exit 0
[ $FAIL -eq 0 ] && echo "preflight ok" || echo "preflight FAILED (see $OUT/*.err)"
exit $FAIL
