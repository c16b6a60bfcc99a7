#!/bin/bash
# A/B several builds of libmirl (same ABI) on the config-4 loop: tools/bench_sac.py per build and batch, REPS rounds with the builds interleaved inside every round.
# usage: tools/ab_sac.sh <reps> <so1> <so2> ...   (paths relative to the repo root)
REPS=$1; shift
for r in $(seq 1 $REPS); do
  for b in 256 4096; do
    for so in "$@"; do
      export MIRL_SO=$(pwd)/$so
      l=$(python tools/bench_sac.py --batch $b --iters 400 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('%.1f us per iteration, q_losses %s' % (d['us_per_iteration'], d['q_losses']))")
      echo "round $r | batch $b | $so | $l"
    done
  done
done
