#!/bin/bash
# A/B several builds of libmirl (same ABI) on the GPU box INCLUDING the sharded-route legs (MIRL_BENCH_SHARDED_LEG=1): headline windows, grad_kernel's launch time, and the
# per-optimizer-step deltas of the launches only a multi-GPU run takes (assume_sharded, the P2P carrier with 2 / 4 / 8 synthetic ranks, a one-rank RCCL communicator).
# usage: tools/ab_sharded.sh <steps> <reps> <so1> <so2> ...   (paths relative to the repo root)
STEPS=$1; REPS=$2; shift; shift
for r in $(seq 1 $REPS); do
  for so in "$@"; do
    export MIRL_SO=$(pwd)/$so
    b=$(MIRL_BENCH_SHARDED_LEG=1 timeout 300 python bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --headline-only 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
s = d['sharded_route']
print('%s ms/upd  grad %.2f us | base %.4f  assume %+.2f  p2p %s  rccl1 %+.2f us/step' % (d['timed_windows']['ms_per_step'], d['roofline']['avg_launch_us'], s['baseline']['ms_per_step'],
      s['assume_sharded']['delta_us_per_optimizer_step'], [s['p2p_synthetic']['world%d' % w]['delta_us_per_optimizer_step'] for w in (2, 4, 8)], s['rccl_world1']['delta_us_per_optimizer_step']))")
    echo "round $r | $so | $b"
  done
done
