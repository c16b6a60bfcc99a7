#!/bin/bash
# A/B several builds of libmirl (same ABI) on the GPU box: `bench.py --headline-only --no-cpu-baseline` per build, REPS rounds interleaved (build order inside every round),
# one line per run: ms per update (three windows), grad_kernel's HIP-event launch time, kernel breakdown.
# usage: tools/ab_headline.sh <steps> <reps> <so1> <so2> ...   (paths relative to the repo root)
STEPS=$1; REPS=$2; shift; shift
for r in $(seq 1 $REPS); do
  for so in "$@"; do
    export MIRL_SO=$(pwd)/$so
    b=$(timeout 300 python bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --headline-only 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('%s ms/upd  grad %.2f us (%.4f of peak)  %s' % (d['timed_windows']['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], {k: v for k, v in d['kernel_ms_per_update_bracketed'].items() if k != 'note'}))")
    echo "round $r | $so | $b"
  done
done
